#!/bin/bash
# Runs ON THE GPU BOX: Encodec parity tests, then kernel trace + timeline of the C3 step (tag = $1)
cd $GRAFT_REPO_ROOT
TAG=${1:-r6b}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 900 python -m pytest tests/test_encodec_gpu.py tests/test_baseline_sizes_gpu.py::test_c3_encodec48k_batch16x2s_vs_oracle_and_batch_invariance -m gpu -q -x > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log
python tools/codecbench.py --only encodec48 --steps 20 --warmup 5 --classes 2>&1 | tail -3 | cut -c1-1500
bash tools/probe/prof_enc.sh $TAG
cp gpurun_out/prof_$TAG/${TAG}_encodec48.* $OUT/ 2>/dev/null
