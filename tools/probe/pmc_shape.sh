#!/bin/bash
# Runs ON THE GPU BOX: counters (one PMC pass per call, kernel trace only) of single conv shapes through tools/probe/shapebench.py
#   [PMC="CTR1 CTR2 ..."] bash tools/probe/pmc_shape.sh <tag> <spec> [<spec> ...]      prints per-launch sums of each counter
TAG=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
PMC=${PMC:-SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES}
i=0
for spec in "$@"; do
  i=$((i+1))
  rocprofv3 --pmc $PMC --kernel-trace -d $OUT/p$i -o p --output-format csv -- python3 $R/tools/probe/shapebench.py $spec > $OUT/p$i.log 2>&1
  python3 - $OUT/p$i "$spec" <<'PY'
import csv, glob, sys, collections
d, spec = sys.argv[1], sys.argv[2]
f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for row in csv.DictReader(open(f[0])):
    k = row['Kernel_Name'][:48]
    acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    n[k].add(row['Dispatch_Id'])
print('##', spec)
for k, c in acc.items():
    if 'conv' not in k: continue
    print('  ', k, 'launches', len(n[k]), ' '.join(f"{name}={v/len(n[k]):.4g}" for name, v in sorted(c.items())))
PY
  rm -rf $OUT/p$i
done
