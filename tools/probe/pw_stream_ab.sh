#!/bin/bash
# Runs ON THE GPU BOX: the streaming pointwise kernel against the tile-per-workgroup one on the narrow long rows (steady state:
# tools/probe/clockshape.py loops each shape for ~250 ms)
cd $GRAFT_REPO_ROOT
S="8,192,192,1,1,0,110592,0,4 32,192,192,1,1,0,22272,0,4 8,192,192,1,1,0,110592,0,6 8,96,96,1,1,0,221184,0,4 8,128,128,1,1,0,110592,0,4 8,64,64,1,1,0,221184,0,4"
echo "== NC_PW_STREAM=1"; NC_PW_STREAM=1 python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids
echo "== default (tile-per-workgroup kernel)"; python tools/probe/clockshape.py $S 2>&1 | grep -v amdgpu.ids
