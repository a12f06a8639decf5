cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/keep; mv build_abl/lib_traceprio3.so gpurun_out/keep/ 2>/dev/null
bash tools/probe/ab_libs3.sh > /dev/null 2>&1
mv gpurun_out/keep/lib_traceprio3.so build_abl/
cat gpurun_out/ab_libs3.txt
sed -i 's/for lib in build_abl\/lib_trace\*.so; do/for lib in build_abl\/lib_traceprio3.so; do/' tools/probe/run_conv_trace.sh
bash tools/probe/run_conv_trace.sh > /dev/null 2>&1; cat gpurun_out/conv_trace.txt
bash tools/probe/tm_pick_up.sh 2>&1 | head -9
