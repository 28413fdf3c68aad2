#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
PROBE_TAG=default python tools/d2h_probe.py
PROBE_TAG=sdma0 HSA_ENABLE_SDMA=0 python tools/d2h_probe.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $R/gpurun_out/h_probe -- python3 $R/tools/d2h_probe.py > $R/gpurun_out/h_probe.log 2>&1
cd $R
head -5 gpurun_out/h_probe/*/*kernel_stats.csv | cut -c1-150
ls gpurun_out/h_probe/*/ | head; head -5 gpurun_out/h_probe/*/*memory_copy_stats.csv 2>/dev/null | cut -c1-200
rm -f gpurun_out/h_probe/*/*trace.csv
