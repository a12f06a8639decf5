#!/bin/bash
# Runs ON THE GPU BOX: row-tile height of the pointwise kernel on the wide layers (NC_TM_PICK forces a packed variant where it exists)
cd $GRAFT_REPO_ROOT
S="8,384,384,1,1,0,36864,0,4 8,768,768,1,1,0,4608,0,4 8,256,256,1,1,0,36864,0,4 32,384,384,1,1,0,5568,0,4 32,768,768,1,1,0,696,0,4 32,512,512,1,1,0,696,0,4"
for tm in 0 4 3 2; do echo "== NC_TM_PICK=$tm"; NC_TM_PICK=$tm python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids; done
