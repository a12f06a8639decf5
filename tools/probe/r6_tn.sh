#!/bin/bash
# Runs ON THE GPU BOX: the 696-step k = 7 layers of DAC C2 (C = 512 / 768; fuse = 3: Snake in + Snake out) with 256- vs 128-column tiles
cd $GRAFT_REPO_ROOT
S="32,768,768,7,1,3,696,0,3 32,512,512,7,1,3,696,0,3"
for tn in 192 700; do for tm in 0 2 3 4; do echo "== NC_TN_THRESH=$tn NC_TM_PICK=$tm"; NC_TN_THRESH=$tn NC_TM_PICK=$tm python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids; done; done
