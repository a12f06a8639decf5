#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3i; mkdir -p $OUT
timeout 900 python -m pytest tests/test_encodec_gpu.py tests/test_baseline_sizes_gpu.py tests/test_containers_gpu.py -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | head -5
for i in 1 2 3; do timeout 300 python tools/codecbench.py --only encodec48 --steps 20 --warmup 3 2>/dev/null | tail -1; done
