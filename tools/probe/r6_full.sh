#!/bin/bash
# Runs ON THE GPU BOX: same-box A/B of the round-6 Encodec changes (codecbench, alternating), then the whole -m gpu suite + smoke
cd $GRAFT_REPO_ROOT
TAG=${1:-r6d}; OUT=gpurun_out/$TAG; mkdir -p $OUT
for rep in 1 2 3; do
  for v in "default" "NC_NO_RES_A=1" "NC_RMS_TWO_PASS=1" "NC_NO_RES_A=1 NC_RMS_TWO_PASS=1"; do
    ms=$(env $( [ "$v" = default ] || echo $v ) python tools/codecbench.py --only encodec48 --steps 30 --warmup 5 2>/dev/null | grep -o '"ms": [0-9.]*' | head -1)
    echo "$rep | $v | $ms" | tee -a $OUT/ab_encodec.txt
  done
done
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
