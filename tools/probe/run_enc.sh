#!/bin/bash
# Runs ON THE GPU BOX: Encodec GPU tests + the C3 step through tools/codecbench.py; logs under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-enc}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_encodec_gpu.py tests/test_baseline_sizes_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | head -8
timeout 600 python tools/codecbench.py --only encodec --classes --steps 10 > $OUT/codecbench.log 2>&1; tail -3 $OUT/codecbench.log | python -c "import sys,json; [print(k, {a:b for a,b in v.items() if a!=\"classes\"}, *[\"\\n     %s %s\" % kv for kv in v.get(\"classes\",{}).items()]) for k,v in json.loads(sys.stdin.read().strip().splitlines()[-1]).items()]"
