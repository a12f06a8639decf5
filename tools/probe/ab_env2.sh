#!/bin/bash
# Runs ON THE GPU BOX: headline class table A/B of environment settings on one box, interleaved, $REPS rounds.
#   tools/probe/ab_env2.sh "NC_NO_DEEP=1" ""        (each argument = one setting; "" = default)
cd $GRAFT_REPO_ROOT
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
for setting in "$@"; do
  env $setting python bench.py --no-cpu-baseline --no-extra --no-check --steps ${STEPS:-10} --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$setting]', d['ms_per_step'], d['ms_per_step_median'], {k: round(v['ms_per_step'],3) for k, v in d['roofline']['all_classes'].items()})"
done
done
