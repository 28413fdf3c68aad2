#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/q_tests.log 2>&1; rc=$?; echo "parity rc=$rc"; tail -3 gpurun_out/q_tests.log
[ $rc -eq 0 ] || exit 1
for k in 1 2; do
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/q_bench_$k.json 2> gpurun_out/q_bench_$k.err; echo "bench rc=$?"
done
python - <<'P'
import json
for f in ('q_bench_1','q_bench_2'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, round(d['ms_per_step'],2), round(d.get('ms_per_step_one_pass_at_a_time',0),2), {k:round(v,2) for k,v in d.get('stage_ms',{}).items() if k.startswith('tree')})
P
