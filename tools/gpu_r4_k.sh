#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/k_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/k_tests.log
timeout -k 10 120 python tools/fuzz_gpu.py 60 x 999 > gpurun_out/k_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/k_fuzz.log | cut -c1-200
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/k_bench1.json 2> gpurun_out/k_bench1.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/k_bench1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['ms_per_step_one_pass_at_a_time'], {k:round(v,2) for k,v in d['stage_ms'].items() if v>0.3})
P
