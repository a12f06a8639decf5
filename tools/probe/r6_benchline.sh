#!/bin/bash
# Runs ON THE GPU BOX: the new odd-length Encodec tests, then the default bench line (with profiles/traffic.json of this build in place)
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_final; mkdir -p $O
true
timeout 900 python bench.py > $O/r06_bench_full.json 2> $O/r06_bench_full.err; echo bench rc=$?
cp gpurun_out/bench_detail.json $O/r06_bench_detail.json
python -c "
import json
l=[x for x in open('$O/r06_bench_full.json') if x.startswith('{')][-1]
d=json.loads(l); print(len(l), d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline']['traffic_stale'], d['roofline']['traffic_over_algorithmic'], d['roofline']['mfma_busy'], d['roofline']['valu_busy'], d['roofline']['pipe_busy'])
print({k:v for k,v in d.items() if k.startswith('c') and k.endswith('ms_per_step')}, d['cpu_baseline']['value'], d['cpu_baseline']['aten_proxy'], d['encode_only'], d['decode_only'])
"
