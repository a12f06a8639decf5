#!/bin/bash
# Runs ON THE GPU BOX: the parity suites under each fallback switch (the non-default code paths stay correct).  A subset of this matrix
# runs inside `pytest -m gpu` (tests/test_children_gpu.py::test_parity_under_fallback_switches).
# The hand-picked rows below combine switches that do not mask each other; `envmatrix.sh --each` instead ENUMERATES the engine's one
# switch table (nc_debug_switches(), csrc/nc_util.hip) and runs the suites once per boolean switch, so a new switch is covered the
# day it is added to the table.
# Round 5: the switches of measured-and-rejected kernels (docs starting "EXPERIMENTS=1 builds") exist in libnc_mi355x_exp.so only
# (`make -C neuralcodecs_amd/csrc EXPERIMENTS=1`, built HERE when missing -- the one matrix row that builds it); rows naming one run with
# NC_MI355X_LIB pointing at that library.
cd $GRAFT_REPO_ROOT
EXP=$PWD/neuralcodecs_amd/libnc_mi355x_exp.so
expl() { [ -f $EXP ] || make -C neuralcodecs_amd/csrc EXPERIMENTS=1 -j16 -s >/dev/null 2>&1; run NC_MI355X_LIB=$EXP "$@"; }
run() { echo "== $*"; env "$@" timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_encodec_gpu.py tests/test_snac_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -3; }
if [ "$1" = "--each" ]; then
  for sw in $(python - <<'PY'
from neuralcodecs_amd import _lib
for line in _lib.lib().nc_debug_switches().decode().splitlines():
    name, kind, _ = line.split("\t")
    if kind in "bp" and name not in ("NC_LSTM_FAKE_TIMEOUT",):
        print(("X:" if _.startswith("EXPERIMENTS=1") else "") + name)
PY
); do case $sw in X:*) expl ${sw#X:}=1;; *) run $sw=1;; esac; done
  exit 0
fi
run NC_DEFAULT=1
run NC_NO_FLAT=1
run NC_CO_GROUP=1
run NC_LSTM_CHUNKS=1 NC_EUCLID_NO_MFMA=1 NC_THIN_NO_VEC=1
run NC_NO_WIDE_FUSE=1 NC_NO_TILE_ALTS=1
run NC_WIDE_FUSE_192=1 NC_LSTM_CHUNKS=7 NC_LSTM_EVEN_CHUNKS=1
run NC_NO_GN_FUSE=1 NC_NO_IN2=1 NC_NO_CONV3S=1
run NC_NO_GN_FINISH=1 NC_NO_THIN_INM=1 NC_DW_NO_VEC=1
run NC_NO_FLAT_GN=1 NC_LSTM_NO_ELU=1 NC_NO_DIST_SMALL=1 NC_LSTM_UB=2 NC_NO_SUBPIXEL_ANY=1
run NC_NO_CONV_SMALL=1
run NC_SMALL_ROLLED=1 NC_SMALL_MAX_GRID=100000
run NC_SMALL_TN=1 NC_SMALL_WIDE_BELOW=100000 NC_SMALL_K1_COLS=0
run NC_LSTM_STEPWISE=1 NC_NO_TINY_TILES=1 NC_NO_SUBPIXEL=1
run NC_NO_FUSE=1 NC_ENCODEC_NO_FUSE=1 NC_DAC_RVQ_STAGEWISE=1
# round 4
run NC_SNAC_NO_FUSE=1 NC_ATTN_NO_MFMA=1 NC_LN_TILE=0
run NC_SNAC_FUSE_MIN_COLS=0 NC_LN_TILE=16
run NC_SYNC_ACQUIRE=1
run NC_NO_XR=1
run NC_NO_XV=1
run NC_NO_XV_K7=1
# round 6: the streaming Encodec kernels and their layouts back on the paths they replaced; the DAC row pitch (opt-in, measured slower)
run NC_NO_RES_A=1 NC_NO_DOWN2=1 NC_NO_DOWN4=1 NC_NO_DOWN5=1 NC_NO_UP2=1 NC_NO_UP4=1
run NC_NO_UP_PITCH=1 NC_RMS_TWO_PASS=1 NC_LSTM_CHUNKS=3
run NC_LSTM_NO_HTILE=1
run NC_DAC_PITCH=1
# round 5: the experiments library (measured-and-rejected kernels)
expl NC_LSTM_FUSED=1 NC_RVQ_8WAVES=1
expl NC_LSTM_SPLIT=1
expl NC_PW_STREAM=1
expl NC_DUO=1
