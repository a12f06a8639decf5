cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 150 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit time --pc-sampling-method host_trap --pc-sampling-interval 100 --kernel-trace -d $R/gpurun_out/pcs -o p --output-format csv -- python3 $R/tools/convbench.py --iters 3 --filter "dec.k7 C384 d1" > $R/gpurun_out/pcs.log 2>&1
echo rc=$?
tail -5 $R/gpurun_out/pcs.log
ls -la $R/gpurun_out/pcs | head
