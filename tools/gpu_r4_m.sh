#!/bin/bash
# tile-local list ranking: parity in the forced mode, the headline with and without it, kernel times
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
POVU_HIP_TILE_RANK=2 timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/m_tests_forced.log 2>&1; rc=$?; echo "forced rc=$rc"; tail -3 gpurun_out/m_tests_forced.log
[ $rc -eq 0 ] || exit 1
POVU_HIP_TILE_RANK=2 timeout -k 10 200 python tools/fuzz_gpu.py 40 > gpurun_out/m_fuzz_forced.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/m_fuzz_forced.log
timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/m_bench_tile.json 2> gpurun_out/m_bench_tile.err; echo "bench rc=$?"
POVU_HIP_TILE_RANK=0 timeout -k 10 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/m_bench_walk.json 2> gpurun_out/m_bench_walk.err; echo "bench rc=$?"
python - <<'P'
import json
for f in ('m_bench_tile','m_bench_walk'):
    d=json.loads(open('gpurun_out/%s.json'%f).read().strip().splitlines()[-1])
    print(f, d['ms_per_step'], d.get('ms_per_step_one_pass_at_a_time'), {k:round(v,2) for k,v in d.get('stage_ms',{}).items() if k.startswith('tree')})
P
export TMPDIR=/tmp
rm -rf gpurun_out/m_prof; timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/m_prof -o m -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-overlap --no-latency-leg > gpurun_out/m_prof.log 2>&1; echo "prof rc=$?"
CSV=$(find gpurun_out/m_prof -name "*kernel_stats.csv" | head -1); python tools/kstats.py $CSV 4 60 > gpurun_out/m_kstats.txt; grep -i "total\|tile\|rank\|wyllie" gpurun_out/m_kstats.txt || true
find gpurun_out/m_prof -name "*kernel_trace*" -size +20M -delete
