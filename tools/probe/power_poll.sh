#!/bin/bash
# Runs ON THE GPU BOX: socket power and shader clock while the headline bench loops (rocm-smi polling next to a long bench run).
python bench.py --steps 400 --warmup 3 --no-cpu-baseline > gpurun_out/power_bench.log 2>&1 &
BP=$!
sleep 9
for i in $(seq 1 14); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "sclk|Power" | tr '\n' ' '; echo; sleep 1; done
wait $BP
tail -1 gpurun_out/power_bench.log | cut -c1-160
