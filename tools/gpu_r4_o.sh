#!/bin/bash
# kernel times of `decompose -s` on BASELINE config 4 at full size
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out; export TMPDIR=/tmp
POVU_HIP_SUB_TIMES=1 timeout -k 10 400 python tools/time_sub.py 1e8 > gpurun_out/r04_sub_config4_phases.log 2>&1; echo "phases rc=$?"; tail -14 gpurun_out/r04_sub_config4_phases.log
rm -rf gpurun_out/o_prof
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/o_prof -o o -- python3 tools/time_sub.py 1e8 > gpurun_out/o_prof.log 2>&1; echo "prof rc=$?"
CSV=$(find gpurun_out/o_prof -name "*kernel_stats.csv" | head -1); cp $CSV gpurun_out/r04_sub_config4_kernel_stats.csv
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/r04_sub_config4_kernel_stats.csv')))
for r in sorted(rows,key=lambda r:-int(r['TotalDurationNs'])):
    if 'k_sub_' in r['Name']:
        print(r['Name'].split('k_sub_')[1].split('(')[0], r['Calls'], round(int(r['TotalDurationNs'])/2e6,2), 'ms/pass')
P
find gpurun_out/o_prof -name "*kernel_trace*" -delete
