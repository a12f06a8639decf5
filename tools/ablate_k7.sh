#!/bin/bash
# Ablation / trace builds of the k=7 convolution kernel (the measurements quoted in DESIGN.md section 8).
#
#   tools/ablate_k7.sh build      (in the build container)  -> build_abl/lib_<variant>.so, one per macro set below
#   tools/ablate_k7.sh run        (on the GPU box, e.g. `gpurun -- bash tools/ablate_k7.sh run`): swaps each library in and
#                                 times the k=7 layer shapes of the headline config with tools/convbench.py
#   tools/ablate_k7.sh trace      (GPU box) per-wave phase timeline of one CU: tools/probe/convtrace.py + tracesum.py + tracesteps.py
#
# Variants: NC_ABL_NOFRAG (no LDS fragment reads), NC_ABL_NOSTAGE (no staging of the next reduction block), NC_ABL_NOBAR,
# NC_ABL_NOSNAKE, NC_ABL_NOLOADA / NOLOADX / NOSTOREA (staging sub-steps), NC_NSEG=n (pipeline segments), NC_FRAG_DEPTH=n,
# NC_DBG_TRACE (s_memtime stamps per phase, staged in LDS; needs nc_conv.hip built with the same macro for the LDS size).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=$ROOT/neuralcodecs_amd/csrc
OUT=$ROOT/build_abl
FL="--offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -fvisibility=hidden -DNC_EXP_SCHED"
VARIANTS=("nofrag:-DNC_ABL_NOFRAG" "nostage:-DNC_ABL_NOSTAGE" "nobar:-DNC_ABL_NOBAR" "nosnake:-DNC_ABL_NOSNAKE" "noloada:-DNC_ABL_NOLOADA"
          "noloadx:-DNC_ABL_NOLOADX" "all:-DNC_ABL_NOFRAG -DNC_ABL_NOSTAGE -DNC_ABL_NOBAR")
case "$1" in
build)
    make -C $SRC -j8 -s
    mkdir -p $OUT && cp $ROOT/neuralcodecs_amd/libnc_mi355x.so $OUT/lib_base.so
    OTHERS=$(ls $SRC/build/*.o | grep -v "nc_conv_k7.o")
    for v in "${VARIANTS[@]}"; do
        n=${v%%:*}; d=${v#*:}
        /opt/rocm/bin/hipcc $FL $d -c $SRC/nc_conv_k7.hip -o $OUT/k7_$n.o
        /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_$n.so $OTHERS $OUT/k7_$n.o
    done
    OTHERS=$(ls $SRC/build/*.o | grep -v "nc_conv_k7.o\|nc_conv.o")
    /opt/rocm/bin/hipcc $FL -DNC_DBG_TRACE -c $SRC/nc_conv_k7.hip -o $OUT/k7_trace.o
    /opt/rocm/bin/hipcc $FL -DNC_DBG_TRACE -c $SRC/nc_conv.hip -o $OUT/conv_trace.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/lib_trace.so $OTHERS $OUT/k7_trace.o $OUT/conv_trace.o
    rm -f $OUT/*.o; ls $OUT ;;
run)
    cd $ROOT
    for f in build_abl/lib_*.so; do
        n=$(basename $f .so); [ "$n" = lib_trace ] && continue
        cp $f neuralcodecs_amd/libnc_mi355x.so
        echo "== $n"; python tools/convbench.py --filter "k7 C" 2>&1 | grep -E "C768 d1|C384 d1|C192 d1|C256 d1|C128 d1"
    done ;;
trace)
    cd $ROOT
    cp build_abl/lib_trace.so neuralcodecs_amd/libnc_mi355x.so
    python tools/probe/convtrace.py ${2:-384} ${3:-5568} ${4:-3}
    python tools/probe/tracesum.py gpurun_out/convtrace.npy
    python tools/probe/tracesteps.py gpurun_out/convtrace.npy ;;
*) echo "usage: $0 build|run|trace [C T fuse]"; exit 2 ;;
esac
