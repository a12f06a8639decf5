#!/bin/bash
# Runs ON THE GPU BOX: GPU suite + full default bench line, logs under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
TAG=${1:-run}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | head -8
timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -1 $OUT/bench_full.json | cut -c1-400
