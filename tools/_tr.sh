timeout 300 python tools/convbench.py --filter "k1 C" 2>&1 | tail -12
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python tools/codecbench.py 2>&1 | tail -1
