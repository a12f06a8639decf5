#!/bin/bash
# Runs ON THE GPU BOX: the parity suites under each fallback switch (the non-default code paths stay correct)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_encodec_gpu.py tests/test_snac_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -3; }
run NC_DEFAULT=1
run NC_NO_FLAT=1
run NC_CO_GROUP=1
run NC_LSTM_CHUNKS=1 NC_EUCLID_NO_MFMA=1 NC_THIN_NO_VEC=1
run NC_NO_WIDE_FUSE=1 NC_NO_TILE_ALTS=1
run NC_WIDE_FUSE_192=1 NC_LSTM_CHUNKS=7
