#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3e; mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q > $OUT/pytest_ops.log 2>&1; grep -E "passed|failed|Error|error" $OUT/pytest_ops.log | head -5
timeout 300 python tools/convbench.py --iters 10 --filter ru 2>&1 | grep res_unit
sumline() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"; }
for rep in 1 2; do for g in 1 0; do echo "no_wide_fuse=$g"; NC_NO_WIDE_FUSE=$g timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline; done; done
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log
