#!/bin/bash
# Runs ON THE GPU BOX: headline bench A/B of build_abl/lib_*.so on one box, interleaved, 3 rounds (class table per run)
cd $GRAFT_REPO_ROOT
cp neuralcodecs_amd/libnc_mi355x.so /tmp/orig.so
for rep in 1 2 3; do
for f in build_abl/lib_*.so; do
  v=$(basename $f .so)
  cp $f neuralcodecs_amd/libnc_mi355x.so
  python bench.py --no-cpu-baseline --no-extra --no-check --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'], {k: round(v['ms_per_step'],2) for k, v in d['roofline']['all_classes'].items()})"
done
done
cp /tmp/orig.so neuralcodecs_amd/libnc_mi355x.so
