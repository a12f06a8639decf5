#!/bin/bash
# Runs ON THE GPU BOX: A/B of build_abl/lib_*.so (tools/probe/mk_abl.sh or copies of full builds) over the DAC headline step and the
# other configs, interleaved, $REPS rounds.  Variants are selected through NC_MI355X_LIB: the shipped library is never overwritten.
cd $GRAFT_REPO_ROOT
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
for f in build_abl/lib_*.so; do
  v=$(basename $f .so)
  NC_MI355X_LIB=$PWD/$f python bench.py --no-cpu-baseline --no-extra --no-check --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v dac', d['ms_per_step'], {k: round(v['ms_per_step'],2) for k, v in d['roofline']['all_classes'].items() if k in ('conv_k7','conv_k1','conv_up','conv_down')})"
  NC_MI355X_LIB=$PWD/$f python tools/codecbench.py --classes --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', {k: v['ms'] for k, v in d.items()}, 'snac44', {c: round(x['ms'],2) for c,x in d['snac44k_c5_share']['classes'].items() if c in ('conv_k1','conv_up','conv_down')}, 'enc48', {c: round(x['ms'],2) for c,x in d['encodec48k_c3']['classes'].items() if c in ('conv_k1','conv_up','conv_down','conv_misc')})"
done
done
