#!/bin/bash
cd $GRAFT_REPO_ROOT
cp neuralcodecs_amd/libnc_mi355x.so /tmp/orig.so
sumline() { python -c "import json,sys; d=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1]); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], {k: v['ms_per_step'] for k, v in d['roofline']['all_classes'].items()})"; }
for rep in 1 2; do
for f in build_abl/lib_a_275f.so build_abl/lib_b_cur.so; do
  cp $f neuralcodecs_amd/libnc_mi355x.so
  echo "== $f"; timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline
  if [ $f = build_abl/lib_b_cur.so ]; then echo "-- no wide fuse"; NC_NO_WIDE_FUSE=1 timeout 300 python bench.py --no-cpu-baseline --no-extra 2>/dev/null | sumline; fi
done
done
cp build_abl/lib_b_cur.so neuralcodecs_amd/libnc_mi355x.so
timeout 300 python tools/convbench.py --iters 10 --filter ru 2>&1 | grep res_unit
timeout 300 python tools/codecbench.py --only encodec48 --steps 10 --warmup 3 2>/dev/null | tail -1
cp /tmp/orig.so neuralcodecs_amd/libnc_mi355x.so
