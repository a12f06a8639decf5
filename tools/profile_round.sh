#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): rocprofv3 kernel trace and the two HBM-traffic PMC passes of the headline bench command.
# Outputs land under gpurun_out/prof_$1/ ; summarise afterwards with tools/rocpd_summary.py, tools/pmc_summary.py, tools/traffic_json.py.
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o dac_b32 -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/pmc_fetch -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/pmc_write -o p --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1
find $OUT -type f | head -30
tail -2 $OUT/trace.log | cut -c1-300
