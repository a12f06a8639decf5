#!/bin/bash
# Runs ON THE GPU BOX: the round's final evidence on the FINAL build -- full GPU suite + smoke, the default bench line, the parity sweep and a
# short soak, kernel traces + PMC passes (tools/profile_round.sh) and the per-class traffic table bound to the library's hash.
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final; mkdir -p $O
(timeout 2400 python -m pytest tests -m gpu -q -rs > $O/r06_final_pytest.log 2>&1; echo "pytest rc=$?" >> $O/r06_final_pytest.log)
tail -8 $O/r06_final_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > $O/r06_bench_full.json 2> $O/r06_bench_full.err; echo bench rc=$?
cp gpurun_out/bench_detail.json $O/r06_bench_detail.json 2>/dev/null
timeout 1500 python tools/gpu_parity_sweep.py --weight-seeds 42,7,1234 --pcm-seeds 1,2,3 --presets --out $O/r06_gpu_parity_sweep.json > $O/sweep.log 2>&1; tail -2 $O/sweep.log | cut -c1-300
timeout 600 python tools/soak.py --iters 10000 --out $O/r06_soak_default.json > $O/soak.log 2>&1; tail -1 $O/soak.log | cut -c1-300
bash tools/profile_round.sh r06 > $O/profile_round_r06.log 2>&1
P=gpurun_out/prof_r06
LIB=neuralcodecs_amd/libnc_mi355x.so
for wl in dac:dac44k encodec48:encodec48k snac44:snac44k; do
  n=${wl%%:*}; k=${wl#*:}
  python tools/pmc_classes.py --key $k --lib $LIB --out $O/traffic_r06.json \
    fetch=$(find $P/pmc_${n}_fetch -name '*counter_collection.csv' | head -1):$P/launch_${n}_fetch.log \
    write=$(find $P/pmc_${n}_write -name '*counter_collection.csv' | head -1):$P/launch_${n}_write.log \
    sq=$(find $P/pmc_${n}_sq -name '*counter_collection.csv' | head -1):$P/launch_${n}_sq.log 2>&1 | tail -3
done
cp $P/r06_*.kernel_stats.txt $O/ 2>/dev/null
sha256sum $LIB | tee $O/lib.sha256
ls -la $O
rm -rf $P/trace_* $P/pmc_*      # (tens of MB of databases: only the summaries travel back)
