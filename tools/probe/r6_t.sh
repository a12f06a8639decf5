#!/bin/bash
# Runs ON THE GPU BOX: the test files given as arguments (default: the group tests)
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest ${@:-tests/test_group_gpu.py} -m gpu -q 2>&1 | tail -15
