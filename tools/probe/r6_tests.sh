#!/bin/bash
# Runs ON THE GPU BOX: a subset of the -m gpu suite given as arguments (default: the round-6 additions), log under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${TAG:-r6t}; OUT=gpurun_out/$TAG; mkdir -p $OUT
FILES=${@:-tests/test_nonfinite_gpu.py tests/test_group_gpu.py tests/test_baseline_sizes_gpu.py}
timeout 1500 python -m pytest $FILES -m gpu -q > $OUT/pytest.log 2>&1; tail -25 $OUT/pytest.log
