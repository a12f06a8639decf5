#!/bin/bash
# Full experimental build of the engine with extra compiler flags into build_abl/lib_<name>.so (own object directory; the shipped
# library and build/ are not touched):   tools/probe/mk_full.sh <name> "<extra flags>"
set -e
cd "$(dirname "$0")/../../neuralcodecs_amd/csrc"
name=$1; flags=$2
od=../../build_abl/obj_$name; mkdir -p $od
srcs=$(make -pn 2>/dev/null | grep -E "^SRCS = " | head -1 | sed 's/SRCS = //')
for f in $srcs; do echo "$f"; done | xargs -P 8 -I{} sh -c "/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -Wall -Wno-unused-result $flags -c {} -o $od/\$(basename {} .hip).o 2>&1 | grep -E 'error' || true"
n=$(ls $od/*.o | wc -l); want=$(echo $srcs | wc -w)
[ "$n" = "$want" ] || { echo "BUILD FAILED: $n of $want objects"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/lib_$name.so $od/*.o -ldl
echo built build_abl/lib_$name.so
