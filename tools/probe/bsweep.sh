#!/bin/bash
# Runs ON THE GPU BOX: batch sweep of one conv shape (Cin,Cout,K,stride,pad,T[,tr[,fuse]]) -- how the launch time steps with the grid
cd $GRAFT_REPO_ROOT
SPEC=$1; shift
for b in "$@"; do python tools/probe/shapebench.py $b,$SPEC 2>&1 | grep -v amdgpu.ids; done
