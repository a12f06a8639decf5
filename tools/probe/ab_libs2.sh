#!/bin/bash
# Runs ON THE GPU BOX: headline bench A/B of build_abl/lib_*.so on one box, interleaved, 3 rounds (class table per run).
# The variants are selected through NC_MI355X_LIB (neuralcodecs_amd/_lib.py): the shipped library is never overwritten, so an
# interrupted run cannot leave an experimental build installed.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for f in build_abl/lib_*.so; do
  v=$(basename $f .so)
  NC_MI355X_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-extra --no-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k: round(v['ms_per_step'],2) for k, v in d['roofline']['all_classes'].items()})"
done
done
