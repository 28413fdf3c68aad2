#!/bin/bash
# One GPU-box session: parity tests, the default bench line, and a rocprofv3 kernel trace of the headline config.
# Usage (through gpurun): bash tools/gpu_session.sh <tag> [tests|notests]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-s}
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd $R
if [ "${2:-tests}" = "tests" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -30 $O/pytest.log; exit 1; }
  tail -3 $O/pytest.log
fi
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 2; }
cat $O/bench.json
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --no-secondary --steps 3 --warmup 1 > $O/bench_under_rocprof.json 2> $O/kt.err || { tail -20 $O/kt.err; exit 3; }
KS=$(ls $O/kt/*/*kernel_stats.csv | tail -1)
cp $KS $O/kernel_stats.csv
python3 $R/tools/trace_timeline.py $(ls $O/kt/*/*kernel_trace.csv | tail -1) > $O/timeline.txt 2>/dev/null; tail -1 $O/timeline.txt > $O/timeline_summary.txt
rm -f $O/kt/*/*kernel_trace.csv
head -45 $O/kernel_stats.csv | cut -c1-200
