#!/bin/bash
# Runs ON THE GPU BOX: the round's closing evidence -- GPU suite, smoke, the full default bench line, then the profile set.
cd $GRAFT_REPO_ROOT
TAG=${1:-r02}
OUT=gpurun_out/final_$TAG; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | head -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py > $OUT/bench_full.json 2> $OUT/bench_full.err; tail -1 $OUT/bench_full.json | cut -c1-300
bash tools/profile_round.sh $TAG > $OUT/profile_round.log 2>&1; tail -2 $OUT/profile_round.log | cut -c1-200
