#!/bin/bash
# kernel trace of one bench workload: bash tools/trace_workload.sh <tag> <workload> [extra bench args]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; WL=$2; shift 2
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-secondary --no-overlap --no-latency-leg --steps 2 --warmup 1 "$@" > $O/bench.json 2> $O/kt.err || { tail -20 $O/kt.err; exit 3; }
cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/kernel_stats.csv
python3 $R/tools/trace_timeline.py $(ls $O/kt/*/*kernel_trace.csv | tail -1) > $O/timeline.txt 2>&1; rm -f $O/kt/*/*kernel_trace.csv
head -12 $O/kernel_stats.csv | cut -c1-160
