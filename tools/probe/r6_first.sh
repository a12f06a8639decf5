#!/bin/bash
# Runs ON THE GPU BOX (round 6, first call): the new parity tests (non-finite inputs, C4 / C5 global batches, groups), then a kernel trace +
# timeline of the C3 step
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r6a; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_nonfinite_gpu.py tests/test_group_gpu.py tests/test_baseline_sizes_gpu.py -m gpu -q > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log
bash tools/probe/prof_enc.sh r6a
cp gpurun_out/prof_r6a/r6a_encodec48.* $OUT/ 2>/dev/null
