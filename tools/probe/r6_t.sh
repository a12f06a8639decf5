#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_nonfinite_gpu.py -m gpu -q 2>&1 | tail -15
