#!/bin/bash
# Runs ON THE GPU BOX: GPU tests, A/B of the flattened column axis (NC_NO_FLAT) and LSTM chunk counts.
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3b; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
sumline() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"; }
for rep in 1 2; do for g in 1 0; do echo "no_flat=$g"; NC_NO_FLAT=$g timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline; done; done
for c in 2 3 4 5; do echo "chunks=$c"; NC_LSTM_CHUNKS=$c timeout 300 python tools/codecbench.py --only encodec48 --steps 10 --warmup 3 2>/dev/null | tail -1; done
