#!/bin/bash
# Runs ON THE GPU BOX: the -m gpu suite + smoke, log under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-tests}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed|rror" $OUT/pytest.log | head -5
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
