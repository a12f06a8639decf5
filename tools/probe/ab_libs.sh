#!/bin/bash
# Runs ON THE GPU BOX: A/B of the library variants under build_abl/ (lib_<name>.so) on the same box, twice each, interleaved:
# k=7 layer timings (tools/convbench.py) and the headline bench line.  Variants are selected through NC_MI355X_LIB
# (neuralcodecs_amd/_lib.py); the shipped library is never overwritten.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in build_abl/lib_*.so; do
  v=$(basename $f .so)
  export NC_MI355X_LIB=$PWD/$f
  echo "== $v"; python tools/convbench.py --filter "${1:-k7 C}" --iters 10 2>&1 | grep -E "${2:-C768 d1|C512 d1|C384 d1|sum}"
  python bench.py --no-cpu-baseline --no-extra 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"
done
done
