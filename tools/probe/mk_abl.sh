#!/bin/bash
# Build an experimental variant of the engine into build_abl/lib_<name>.so without touching the shipped library:
#   tools/probe/mk_abl.sh <name> <source.hip> "<extra compiler flags>"
# (every other object comes from the normal build directory; run `make` first)
set -e
cd "$(dirname "$0")/../../neuralcodecs_amd/csrc"
name=$1; src=$2; flags=$3
mkdir -p ../../build_abl/obj
obj=../../build_abl/obj/${src%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-result $flags -c $src -o $obj
objs=$(ls build/*.o | grep -v "build/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/lib_$name.so $objs $obj -ldl
echo built build_abl/lib_$name.so
