#!/bin/bash
# Runs ON THE GPU BOX: after the tile-cost change -- the up-convolutions of DAC C2 / SNAC C5 share at default, then every bench configuration twice
cd $GRAFT_REPO_ROOT
S="8,1536,768,16,8,4,576,1,1 8,768,384,16,8,4,4608,1,1 32,1536,768,16,8,4,87,1,1 32,768,384,16,8,4,696,1,1 32,384,192,8,4,2,5568,1,1 32,192,96,4,2,1,22272,1,1 8,384,192,6,3,2,36864,1,1 8,192,96,4,2,1,110592,1,1"
python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(d['ms_per_step'], d['roofline']['frac'], {k:v[0] for k,v in d['roofline']['all_classes'].items()}, {k:v for k,v in d.items() if k.startswith('c')and k.endswith('ms_per_step')})
" $1; }
for rep in 1 2; do python bench.py --no-cpu-baseline --no-check --steps 10 --warmup 3 > /tmp/b.json 2>/dev/null; show /tmp/b.json; done
