#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/codecbench.py --only snac44 --steps 10 --warmup 3 --classes 2>/dev/null | tail -1 | cut -c1-900
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/tr -o p -- python3 $GRAFT_REPO_ROOT/tools/codecbench.py --only snac44 --steps 3 --warmup 1 > /tmp/l.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/tr -name 'p_results.db' | head -1) | grep "grid=.*conv_mfma_kernel<.*, 2, 2, 16, 20" | cut -c1-200
