python bench.py --steps 150 --warmup 2 --no-cpu-baseline > gpurun_out/clk_bench.log 2>&1 &
BP=$!
sleep 14
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power|fclk" ; echo ---; sleep 0.7; done
wait $BP
tail -1 gpurun_out/clk_bench.log | cut -c1-220
