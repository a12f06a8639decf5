#!/bin/bash
# Runs ON THE GPU BOX: the parity suites under the switches that route the most layers through the LEGACY conv instances (a short form
# of envmatrix.sh for re-checks after a change to the template's legacy path).
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_encodec_gpu.py tests/test_snac_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -3; }
run NC_NO_XV=1
run NC_NO_XV_K7=1
run NC_NO_XR=1
run NC_NO_FUSE=1 NC_ENCODEC_NO_FUSE=1 NC_DAC_RVQ_STAGEWISE=1
run NC_NO_FLAT=1
run NC_SNAC_NO_FUSE=1 NC_NO_WIDE_FUSE=1 NC_NO_TILE_ALTS=1
