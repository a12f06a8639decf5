#!/bin/bash
# Runs ON THE GPU BOX: row-tile height of the sub-pixel up-convolutions (NC_TM_PICK forces a packed variant where it exists)
cd $GRAFT_REPO_ROOT
S="32,1536,768,16,8,4,87,1,0 32,768,384,16,8,4,696,1,0 32,384,192,8,4,2,5568,1,0 32,192,96,4,2,1,22272,1,0 8,1536,768,16,8,4,576,1,0 8,768,384,16,8,4,4608,1,0 8,192,96,4,2,1,110592,1,0"
for tm in 0 4 3 2; do echo "== NC_TM_PICK=$tm"; NC_TM_PICK=$tm python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids; done
