#!/bin/bash
# one-pass scans: parity, then the headline with and without them
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/p_tests.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 gpurun_out/p_tests.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/p_bench_lb.json 2> gpurun_out/p_bench_lb.err; echo "bench rc=$?"
POVU_HIP_SCAN_LOOKBACK=0 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/p_bench_two.json 2> gpurun_out/p_bench_two.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/p_bench_lb2.json 2> gpurun_out/p_bench_lb2.err; echo "bench rc=$?"
python - <<'P'
import json
for f in ('p_bench_lb','p_bench_two','p_bench_lb2'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['ms_per_step'],2), round(d.get('ms_per_step_one_pass_at_a_time',0),2), {k:round(v,2) for k,v in d.get('stage_ms',{}).items()})
P
