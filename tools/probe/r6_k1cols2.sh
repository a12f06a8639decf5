#!/bin/bash
# Runs ON THE GPU BOX: NC_SMALL_K1_COLS 4096 (default) vs 1024 / 2048 on every bench configuration
cd $GRAFT_REPO_ROOT
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], d['ms_per_step'], {k:v for k,v in d.items() if k.startswith('c')and k.endswith('ms_per_step')})
" $1 $2; }
for rep in 1 2; do for v in 4096 2048 1024; do
  NC_SMALL_K1_COLS=$v python bench.py --no-cpu-baseline --no-check --steps 10 --warmup 3 > /tmp/b_$v.json 2>/dev/null; show /tmp/b_$v.json K1_COLS=$v
done; done
