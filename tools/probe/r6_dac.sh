#!/bin/bash
# Runs ON THE GPU BOX: DAC + Encodec parity tests, then the bench line with / without the row pitch
cd $GRAFT_REPO_ROOT
TAG=${1:-r6c}; OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_dac_gpu.py tests/test_encodec_gpu.py tests/test_baseline_sizes_gpu.py::test_c2_dac44k_batch32_vs_oracle_and_batch_invariance tests/test_baseline_sizes_gpu.py::test_c3_encodec48k_batch16x2s_vs_oracle_and_batch_invariance tests/test_nonfinite_gpu.py -m gpu -q -x > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
show() { python -c "
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], d['ms_per_step'], d['roofline']['frac'], {k:v[0] for k,v in d['roofline']['all_classes'].items()}, {k:v for k,v in d.items() if k.startswith('c')and k.endswith('ms_per_step')}, len(json.dumps(d)))
" $1 $2; }
for rep in 1 2; do
python bench.py --no-cpu-baseline --steps 20 --warmup 5 > $OUT/bench_pitch_$rep.json 2>$OUT/bench_pitch.err; show $OUT/bench_pitch_$rep.json pitch
NC_DAC_NO_PITCH=1 python bench.py --no-cpu-baseline --no-extra --steps 20 --warmup 5 > $OUT/bench_nopitch_$rep.json 2>$OUT/bench_nopitch.err; show $OUT/bench_nopitch_$rep.json nopitch
done
