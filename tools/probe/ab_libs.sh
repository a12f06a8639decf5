#!/bin/bash
# Runs ON THE GPU BOX: A/B of the library variants under build_abl/ (lib_<name>.so) on the same box, twice each, interleaved:
# k=7 layer timings (tools/convbench.py) and the headline bench line.
cd $GRAFT_REPO_ROOT
cp neuralcodecs_amd/libnc_mi355x.so /tmp/orig.so
for rep in 1 2; do
for f in build_abl/lib_*.so; do
  v=$(basename $f .so)
  cp $f neuralcodecs_amd/libnc_mi355x.so
  echo "== $v"; python tools/convbench.py --filter "${1:-k7 C}" --iters 10 2>&1 | grep -E "${2:-C768 d1|C512 d1|C384 d1|sum}"
  python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"
done
done
cp /tmp/orig.so neuralcodecs_amd/libnc_mi355x.so
