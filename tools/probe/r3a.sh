#!/bin/bash
# Runs ON THE GPU BOX: GPU tests, A/B of the LSTM layer pipelining (NC_LSTM_CHUNKS) and of the row-tile groups (NC_CO_GROUP), fetch PMC pass.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3a; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
for c in 1 4 6 8; do echo "chunks=$c"; NC_LSTM_CHUNKS=$c timeout 300 python tools/codecbench.py --only encodec48 --steps 10 --warmup 3 2>/dev/null | tail -1; done
sumline() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"; }
for rep in 1 2; do for g in 1 0; do echo "co_group=$g"; NC_CO_GROUP=$g timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline; done; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export NC_LAUNCH_LOG=$R/$OUT/launch_dac_fetch.log
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $R/$OUT/pmc_dac_fetch -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 2 --warmup 1 > $R/$OUT/pmc_dac_fetch.log 2>&1
unset NC_LAUNCH_LOG
cd $R
python tools/pmc_classes.py --key dac44k --out $OUT/traffic_fetch.json fetch=$(find $OUT/pmc_dac_fetch -name '*counter_collection.csv' | head -1):$OUT/launch_dac_fetch.log 2>&1 | tail -20
