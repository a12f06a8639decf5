R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for w in snac encodec; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$w -o p -- python3 $R/tools/codecbench.py --only $w --steps 3 --warmup 1 > $R/gpurun_out/prof_$w.log 2>&1
done
find $R/gpurun_out/prof_snac $R/gpurun_out/prof_encodec -name "*.db" | head
