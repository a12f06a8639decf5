#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_snac_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -5
sumline() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], {k: (v['ms_per_step'], v['algo_GBps']) for k, v in d['roofline']['all_classes'].items() if k in ('head','stem')})"; }
for v in 1 0 1 0; do if [ $v = 1 ]; then export NC_THIN_NO_VEC=1; else unset NC_THIN_NO_VEC; fi; echo "no_vec=$v"; timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline; done
