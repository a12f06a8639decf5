#!/bin/bash
# Runs ON THE GPU BOX: A/B of environment settings over the DAC headline step and the other configs (codecbench), interleaved.
#   tools/probe/ab_env_all.sh "NC_PW_STREAM=1" ""        (each argument = one setting; "" = default)
cd $GRAFT_REPO_ROOT
REPS=${REPS:-2}
for rep in $(seq 1 $REPS); do
for setting in "$@"; do
  env $setting python bench.py --no-cpu-baseline --no-extra --no-check --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$setting] dac', d['ms_per_step'], d['ms_per_step_median'], round(d['roofline']['all_classes']['conv_k1']['ms_per_step'],3))"
  env $setting python tools/codecbench.py --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$setting]', {k: v['ms'] for k, v in d.items()})"
done
done
