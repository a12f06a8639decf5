#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/r3d; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log
for g in 1 0 1 0; do echo "no_flat=$g"; NC_NO_FLAT=$g timeout 300 python tools/codecbench.py --only encodec48 --steps 10 --warmup 3 2>/dev/null | tail -1; done
for t in 0 2 3 4; do echo "tm_pick=$t"; NC_TM_PICK=$t timeout 300 python tools/convbench.py --iters 10 --filter k7 2>&1 | grep -E "k7|sum" ; done
