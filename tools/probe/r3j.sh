#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3j; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | head -5
( time timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err ) 2>&1 | grep real
tail -1 $OUT/bench_full.json | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['achieved'], d['cpu_baseline'].get('gpu_equals_oracle'))
for k,v in d['extra_configs'].items(): print(k, v['ms_per_step'], v['x_realtime'], v.get('gpu_equals_oracle'))
"
