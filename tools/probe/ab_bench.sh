#!/bin/bash
# Runs ON THE GPU BOX: A/B of the library variants under build_abl/ (lib_<name>.so) on the same box, interleaved: headline bench line.
cd $GRAFT_REPO_ROOT
cp neuralcodecs_amd/libnc_mi355x.so /tmp/orig.so
sumline() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"; }
for rep in 1 2 ${REPS}; do
for f in build_abl/lib_*.so; do
  v=$(basename $f .so)
  cp $f neuralcodecs_amd/libnc_mi355x.so
  echo "== $v"; timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline
  if [ -n "$ENC" ]; then timeout 300 python tools/codecbench.py --only encodec48 --steps 10 --warmup 3 2>/dev/null | tail -1; fi
done
done
cp /tmp/orig.so neuralcodecs_amd/libnc_mi355x.so
