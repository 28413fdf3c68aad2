#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/c_bench1.json 2> gpurun_out/c_bench1.err; echo "bench rc=$?"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-overlap > gpurun_out/c_bench1_sync.json 2> gpurun_out/c_bench1_sync.err; echo "bench sync rc=$?"
POVU_BENCH_ONE_DEVICE=1 timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --scale 0.02 > gpurun_out/c_bench2_small.json 2> gpurun_out/c_bench2.err; echo "bench2 rc=$?"
python - <<'P'
import json
for f in ('c_bench1','c_bench1_sync','c_bench2_small'):
    d=json.loads(open(f'gpurun_out/{f}.json').read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d.get('ms_per_step_one_pass_at_a_time'), d['roofline'].get('pass_latency_ms'), d.get('ms_per_step_whole_job'), d.get('phase_ms'))
P
