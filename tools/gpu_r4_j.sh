#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/j_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/j_tests.log
timeout -k 10 200 python tools/fuzz_gpu.py 90 x 4242 > gpurun_out/j_fuzz.log 2>&1; echo "fuzz rc=$?"; tail -1 gpurun_out/j_fuzz.log
timeout -k 10 400 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/j_bench1.json 2> gpurun_out/j_bench1.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/j_bench1.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['ms_per_step_one_pass_at_a_time'], {k:round(v,2) for k,v in d['stage_ms'].items() if v>0.3})
for k,v in d['secondary'].items(): print(k, round(v['ms_per_step'],3))
print(d['end_to_end'].get('wall_s'), d['end_to_end'].get('host_ms'))
P
rocm-smi --showmeminfo vram 2>/dev/null | head -5
