#!/bin/bash
# Runs ON THE GPU BOX: the strided down-convolutions of DAC C2 / SNAC C5-share with the row-tile height forced (NC_TM_PICK), Snake on the input
cd $GRAFT_REPO_ROOT
S="32,64,128,4,2,1,44544,0,1 32,128,256,8,4,2,22272,0,1 32,256,512,16,8,4,5568,0,1 8,64,128,4,2,1,221184,0,1 8,128,256,6,3,2,110592,0,1 8,256,512,16,8,4,36864,0,1 8,512,1024,16,8,4,4608,0,1"
for tm in 0 2 4; do echo "== NC_TM_PICK=$tm"; NC_TM_PICK=$tm python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids; done
