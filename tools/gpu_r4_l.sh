#!/bin/bash
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_sharded.py tests/test_gpu_parity.py -x -q -k "shard or overlapped or engine or bench or cli or gather" > gpurun_out/l_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/l_tests.log
POVU_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 8 --steps 4 --warmup 1 > gpurun_out/l_bench8_rehearsal.json 2> gpurun_out/l_bench8.err; echo "bench8 rc=$?"
timeout -k 10 500 python tools/shard_probe.py 1.0 8 > gpurun_out/l_shard_probe.json 2> gpurun_out/l_shard_probe.err; echo "probe rc=$?"
python - <<'P'
import json
d=json.loads(open('gpurun_out/l_bench8_rehearsal.json').read().strip().splitlines()[-1])
print('8 ranks one device: resident', d['ms_per_step'], 'whole', d['ms_per_step_whole_job'])
d=json.loads(open('gpurun_out/l_shard_probe.json').read().strip().splitlines()[-1])
print(d['single_gpu_ms']); r=d['runs'][0]
print([(round(s['decompose_ms_back_to_back'],2), round(s['decompose_ms_one_at_a_time'],2)) for s in r['shards']], r['resident_shards'], r['whole_job'])
P
