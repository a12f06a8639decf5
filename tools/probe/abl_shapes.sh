#!/bin/bash
# Runs ON THE GPU BOX: shapebench over every build_abl/lib_*.so (tools/probe/mk_abl.sh) for the given shape specs, 2 rounds interleaved.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for f in build_abl/lib_*.so; do
  echo "== $(basename $f .so)"
  NC_MI355X_LIB=$PWD/$f python tools/probe/shapebench.py "$@" 2>&1 | grep -v amdgpu.ids
done
done
