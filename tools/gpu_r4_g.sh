#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "class or walk or nest or tangled or config5 or config3 or random or dense or hub" > gpurun_out/g_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/g_tests.log
for wl in nest tangled; do
timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/g_$wl.json 2> gpurun_out/g_$wl.err; echo "$wl rc=$?"
done
python - <<'P'
import json
for f in ('nest','tangled'):
    d=json.loads(open(f'gpurun_out/g_{f}.json').read().strip().splitlines()[-1])
    print(f, round(d['ms_per_step'],3), round(d['ms_per_step_one_pass_at_a_time'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v>0.3})
P
touch povu_amd/csrc/hip/tree_kernels.hip
make -C povu_amd/csrc -j16 -s EXTRA=-DPOVU_WALK_STATS 2>&1 | grep -E "error" | head
timeout -k 10 300 python bench.py --workload nest --steps 1 --warmup 1 --no-cpu-baseline --no-secondary --no-latency-leg 2>&1 | grep "^walk" | sort | uniq -c | head -4
