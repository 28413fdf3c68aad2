#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "class or walk or nest or tangled or config5 or config3 or random or dense or hub" > gpurun_out/d_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/d_tests.log
for wl in nest tangled hprc-chr; do
timeout -k 10 300 python bench.py --workload $wl --steps 5 --warmup 2 --no-cpu-baseline --no-secondary > gpurun_out/d_$wl.json 2> gpurun_out/d_$wl.err; echo "$wl rc=$?"
done
python - <<'P'
import json
for f in ('nest','tangled','hprc-chr'):
    d=json.loads(open(f'gpurun_out/d_{f}.json').read().strip().splitlines()[-1])
    print(f, round(d['ms_per_step'],3), round(d['ms_per_step_one_pass_at_a_time'],3), {k:round(v,3) for k,v in d['stage_ms'].items() if v>0.3})
P
