#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3c; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for g in 1 0; do echo "no_flat=$g"; NC_NO_FLAT=$g timeout 300 python tools/convbench.py --iters 10 2>&1 | grep -E "C512|C768|C1024|C1536|k3|dec.in|sum|total" ; done
sumline() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"; }
for rep in 1 2; do for g in 1 0; do echo "no_flat=$g"; NC_NO_FLAT=$g timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline; done; done
for g in 1 0; do echo "no_flat=$g"; NC_NO_FLAT=$g timeout 300 python tools/codecbench.py --only encodec48 --steps 10 --warmup 3 2>/dev/null | tail -1; done
for c in 2 3 5; do echo "chunks=$c"; NC_LSTM_CHUNKS=$c timeout 300 python tools/codecbench.py --only encodec48 --steps 10 --warmup 3 2>/dev/null | tail -1; done
