timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 300 python tools/convbench.py 2>&1 | tail -56 | grep -v res_unit
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
