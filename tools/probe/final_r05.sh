#!/bin/bash
# Runs ON THE GPU BOX: the round's final evidence -- full GPU suite, the default bench line, kernel traces + PMC passes (tools/profile_round.sh)
# and the per-class traffic table (tools/pmc_classes.py) of the final build.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
(timeout 2700 python -m pytest tests -m gpu -q -rs > gpurun_out/r05_final_pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r05_final_pytest.log)
tail -8 gpurun_out/r05_final_pytest.log
timeout 900 python bench.py > gpurun_out/r05_bench_full.json 2> gpurun_out/r05_bench_full.err; echo bench rc=$?
bash tools/profile_round.sh r05 > gpurun_out/profile_round_r05.log 2>&1
P=gpurun_out/prof_r05
for wl in dac:dac44k encodec48:encodec48k snac44:snac44k; do
  n=${wl%%:*}; k=${wl#*:}
  python tools/pmc_classes.py --key $k --out gpurun_out/traffic_r05.json \
    fetch=$(find $P/pmc_${n}_fetch -name '*counter_collection.csv' | head -1):$P/launch_${n}_fetch.log \
    write=$(find $P/pmc_${n}_write -name '*counter_collection.csv' | head -1):$P/launch_${n}_write.log \
    sq=$(find $P/pmc_${n}_sq -name '*counter_collection.csv' | head -1):$P/launch_${n}_sq.log 2>&1 | tail -3
done
cp $P/r05_*.kernel_stats.txt gpurun_out/ 2>/dev/null
ls -la gpurun_out/r05_* gpurun_out/traffic_r05.json
rm -rf $P/trace_* $P/pmc_*      # (tens of MB of databases: only the summaries travel back)
