#!/bin/bash
# Runs ON THE GPU BOX: DAC / Encodec / non-finite parity tests, then the DAC kernel table (the stage-fused quantizer launch) and the bench line
cd $GRAFT_REPO_ROOT
TAG=${1:-r6d}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_dac_gpu.py tests/test_ops_gpu.py tests/test_encodec_gpu.py tests/test_baseline_sizes_gpu.py::test_c2_dac44k_batch32_vs_oracle_and_batch_invariance tests/test_nonfinite_gpu.py tests/test_containers_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>$OUT/bench.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(d['ms_per_step'], d['roofline']['frac'], {k:v for k,v in d.items() if k.startswith('c') and k.endswith('ms_per_step')})"
R=$PWD; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/t -o p -- python3 $R/bench.py --no-cpu-baseline --no-extra --steps 6 --warmup 2 > $OUT/prof.log 2>&1
db=$(find $OUT/t -name 'p_results.db' | head -1)
python3 $R/tools/rocpd_summary.py $db 2>/dev/null | grep "grid=.*dac_rvq_fused" | cut -c1-120
rm -rf $OUT/t
