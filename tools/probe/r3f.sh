#!/bin/bash
cd $GRAFT_REPO_ROOT
cp neuralcodecs_amd/libnc_mi355x.so /tmp/orig.so
sumline() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"; }
for f in build_abl/lib_a_275f.so build_abl/lib_b_cur.so; do
  cp $f neuralcodecs_amd/libnc_mi355x.so
  echo "== $f"; timeout 300 python tools/convbench.py --iters 10 2>&1 | grep -E "down|up|dec.in|enc.out|C256|sum"
  echo "-- no wide fuse"; NC_NO_WIDE_FUSE=1 timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline
  echo "-- no wide fuse, no flat"; NC_NO_FLAT=1 NC_NO_WIDE_FUSE=1 timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline
done
cp /tmp/orig.so neuralcodecs_amd/libnc_mi355x.so
