#!/bin/bash
# Build a variant of the XV-only instances (nc_conv_xv*.hip: seconds per file) into build_abl/lib_<name>.so; every other object comes from
# the normal build directory (run `make` first):   tools/probe/mk_xv.sh <name> "<extra compiler flags>"
set -e
cd "$(dirname "$0")/../../neuralcodecs_amd/csrc"
name=$1; flags=$2
mkdir -p ../../build_abl/obj_$name
objs=$(ls build/*.o | grep -v "build/nc_conv_xv")
for f in nc_conv_xv7.hip nc_conv_xv7f.hip nc_conv_xv2.hip nc_conv_xv2g.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-result $flags -c $f -o ../../build_abl/obj_$name/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/lib_$name.so $objs ../../build_abl/obj_$name/*.o -ldl
echo built build_abl/lib_$name.so
