#!/bin/bash
# Runs ON THE GPU BOX: A/B of environment switches on one box: bench.py (no CPU legs) + the C3 / C5 / C1 step times per setting.
# usage: ab_env.sh tag "VAR=a" "VAR=b" ...
cd $GRAFT_REPO_ROOT
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
for setting in "$@"; do
    echo "== $setting"
    env $setting python bench.py --no-cpu-baseline --no-extra --no-check --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  C2 ms', d['ms_per_step'])"
    env $setting python tools/codecbench.py --steps 10 --warmup 3 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  ', {k: v['ms'] for k, v in d.items()})"
done
