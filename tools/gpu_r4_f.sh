#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
touch povu_amd/csrc/hip/tree_kernels.hip
make -C povu_amd/csrc -j16 -s EXTRA=-DPOVU_WALK_STATS 2>&1 | grep -E "error" | head
timeout -k 10 300 python bench.py --workload nest --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-latency-leg 2>&1 | grep "^walk" | sort | uniq -c | head -4
