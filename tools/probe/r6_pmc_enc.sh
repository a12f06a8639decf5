#!/bin/bash
# Runs ON THE GPU BOX: SQ counters of the C3 step's streaming kernels (one counter set per pass, kernel trace only)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r6pmc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
grep -o "SQ_[A-Z_0-9]*" $OUT/counters.txt | sort -u | tr '\n' ' ' | cut -c1-3000
echo
pass() { # name, counters
  rm -rf $OUT/tr
  rocprofv3 --pmc $2 --kernel-trace -d $OUT/tr -o p --output-format csv -- python3 $R/tools/codecbench.py --only encodec48 --steps 2 --warmup 1 > $OUT/$1.log 2>&1
  f=$(find $OUT/tr -name 'p_counter_collection.csv' | head -1)
  python3 - "$f" "$1" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"]
    if not any(s in k for s in ("res_a_kernel","down2_kernel","conv3_stream","conv1x1_kernel<1","conv1x1_kernel<2")): continue
    if int(r["Grid_Size"])<3008*256: continue
    key=k.split("(")[0][:60]+" g"+r["Grid_Size"]
    acc[key][r["Counter_Name"]]+=float(r["Counter_Value"]); n[(key,r["Counter_Name"])]+=1
for key,c in acc.items():
    print(sys.argv[2], key, {k: round(v/n[(key,k)]) for k,v in c.items()})
PY
}
pass a "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"
pass b "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
pass c "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS"
rm -rf $OUT/tr
