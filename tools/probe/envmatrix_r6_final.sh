#!/bin/bash
# Runs ON THE GPU BOX: the rows of envmatrix.sh that route through the kernels changed after the full matrix of profiles/r06_envmatrix.txt was
# recorded (quantizer launches, LSTM h tile), on the final library; first line = its SHA-256.
cd $GRAFT_REPO_ROOT
sha256sum neuralcodecs_amd/libnc_mi355x.so | cut -c1-64
EXP=$PWD/neuralcodecs_amd/libnc_mi355x_exp.so
run() { echo "== $*"; env "$@" timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_dac_gpu.py tests/test_encodec_gpu.py tests/test_snac_gpu.py tests/test_nonfinite_gpu.py -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | head -3; }
run NC_DEFAULT=1
run NC_LSTM_NO_HTILE=1
run NC_LSTM_NO_HTILE=1 NC_SYNC_ACQUIRE=1 NC_LSTM_UB=2
run NC_SYNC_ACQUIRE=1 NC_LSTM_CHUNKS=1 NC_EUCLID_NO_MFMA=1
run NC_LSTM_UB=2 NC_LSTM_CHUNKS=3 NC_DAC_RVQ_STAGEWISE=1
run NC_LSTM_STEPWISE=1 NC_NO_FUSE=1 NC_ENCODEC_NO_FUSE=1
run NC_MI355X_LIB=$EXP NC_RVQ_8WAVES=1
run NC_MI355X_LIB=$EXP NC_LSTM_FUSED=1
