#!/bin/bash
# Runs ON THE GPU BOX: tile sweep (NC_TM_PICK) over the deep Encodec layers, plain and with GroupNorm sums
cd $GRAFT_REPO_ROOT
TAG=${1:-enc_tiles}
OUT=gpurun_out/$TAG; mkdir -p $OUT
P="32,32,64,4,2,1,48000,0 32,64,128,8,4,2,24000,0 32,128,256,10,5,3,6000,0 32,256,512,16,8,4,1200,0 32,512,128,7,1,3,150,0 32,128,512,7,1,3,150,0 32,512,256,16,8,0,150,1 32,256,128,10,5,0,1200,1 32,128,64,8,4,0,6000,1 32,64,32,4,2,0,24000,1"
G="32,32,64,4,2,1,48000,0,8 32,64,128,8,4,2,24000,0,8 32,128,256,10,5,3,6000,0,8 32,256,512,16,8,4,1200,0,8 32,512,128,7,1,3,150,0,8 32,128,512,7,1,3,150,0,8 32,512,256,16,8,0,150,1,8 32,128,64,8,4,0,6000,1,8 32,64,32,4,2,0,24000,1,8"
{
for tm in 0 1 2 3 4; do
  echo "== plain TM_PICK=$tm"; NC_TM_PICK=$tm python tools/probe/shapebench.py $P 2>&1 | grep -v amdgpu.ids
  echo "== gn TM_PICK=$tm"; NC_TM_PICK=$tm python tools/probe/shapebench.py $G 2>&1 | grep -v amdgpu.ids
done
echo "== gn TN_THRESH=100"; NC_TN_THRESH=100 python tools/probe/shapebench.py $G 2>&1 | grep -v amdgpu.ids
echo "== gn TN_THRESH=100000"; NC_TN_THRESH=100000 python tools/probe/shapebench.py $G 2>&1 | grep -v amdgpu.ids
} > $OUT/tiles.log 2>&1
cat $OUT/tiles.log
