#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/i_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/i_tests.log
timeout -k 10 200 python tools/fuzz_gpu.py 120 x 777 > gpurun_out/i_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/i_fuzz.log
timeout -k 10 200 python tools/redo_cost.py > gpurun_out/i_redo_cost.log 2>&1; cat gpurun_out/i_redo_cost.log
