#!/bin/bash
# the inserting passes of -s against the oracle, then the whole GPU suite
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_gpu_subflubbles.py tests/test_gpu_leaf_subflubbles.py -x -q --durations=8 > gpurun_out/n_tests.log 2>&1; rc=$?; echo "sub rc=$rc"; tail -40 gpurun_out/n_tests.log
