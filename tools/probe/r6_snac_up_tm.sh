#!/bin/bash
# Runs ON THE GPU BOX: SNAC 44.1 kHz (C5 share, 8 clips) and DAC C2 stride-8 up-convolutions with the row-tile height forced (NC_TM_PICK); Snake on the input
cd $GRAFT_REPO_ROOT
S="8,1536,768,16,8,4,576,1,1 8,768,384,16,8,4,4608,1,1 32,1536,768,16,8,4,87,1,1 32,768,384,16,8,4,696,1,1"
for tm in 0 4 3 2; do echo "== NC_TM_PICK=$tm"; NC_TM_PICK=$tm python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids; done
