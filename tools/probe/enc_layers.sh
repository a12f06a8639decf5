#!/bin/bash
# Runs ON THE GPU BOX: the deep strided / transposed / k7 layers of the Encodec 48 kHz C3 step (32 segment rows) one by one through
# nc_op_conv1d_bench: plain (flat tiles allowed), one-clip tiles (NC_NO_FLAT), with the GroupNorm sums in the epilogue (fuse 8), and in
# the input mode (fuse 16) / both (24)
cd $GRAFT_REPO_ROOT
TAG=${1:-enc_layers}
OUT=gpurun_out/$TAG; mkdir -p $OUT
DOWN="32,32,64,4,2,1,48000 32,64,128,8,4,2,24000 32,128,256,10,5,3,6000 32,256,512,16,8,4,1200"
UP="32,512,256,16,8,0,150,1 32,256,128,10,5,0,1200,1 32,128,64,8,4,0,6000,1 32,64,32,4,2,0,24000,1"
K7="32,512,128,7,1,3,150 32,128,512,7,1,3,150"
with() { f=$1; shift; for s in "$@"; do echo -n "$s,$f "; done; }
with0() { for s in "$@"; do case $s in *,1) echo -n "$s " ;; *) echo -n "$s,0 " ;; esac; done; }
{
echo "== plain"; python tools/probe/shapebench.py $DOWN $UP $K7
echo "== NC_NO_FLAT"; NC_NO_FLAT=1 python tools/probe/shapebench.py $DOWN $UP $K7
echo "== gn sums (8)"; python tools/probe/shapebench.py $(with 8 $(with0 $DOWN $K7)) $(with 8 $UP)
echo "== input mode (16)"; python tools/probe/shapebench.py $(with 16 $(with0 $DOWN $K7)) $(with 16 $UP)
echo "== both (24)"; python tools/probe/shapebench.py $(with 24 $(with0 $DOWN $K7)) $(with 24 $UP)
} > $OUT/layers.log 2>&1
cat $OUT/layers.log
