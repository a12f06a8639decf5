#!/bin/bash
# Runs ON THE GPU BOX: the whole -m gpu suite + smoke
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/${1:-r6s}; mkdir -p $OUT
timeout 2000 python -m pytest tests -m gpu -q -rs > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
