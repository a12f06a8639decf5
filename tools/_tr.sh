timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
timeout 200 python tools/codecbench.py 2>&1 | tail -1
cp build_abl/lib_trace.so neuralcodecs_amd/libnc_mi355x.so
for f in 3; do echo "fuse=$f"; python tools/probe/convtrace.py 384 5568 $f; python tools/probe/tracesum.py gpurun_out/convtrace.npy; python tools/probe/tracesteps.py gpurun_out/convtrace.npy; done
