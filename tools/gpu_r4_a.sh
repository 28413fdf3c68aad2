#!/bin/bash
# round 4, first GPU session: the whole GPU suite on the new multi-GPU paths + baselines on this box
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/a_tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/a_tests.log
tail -5 gpurun_out/a_tests.log
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary > gpurun_out/a_bench1.json 2> gpurun_out/a_bench1.err; echo "bench rc=$?"
POVU_BENCH_ONE_DEVICE=1 timeout -k 10 400 python bench.py --gpus 8 --steps 3 --warmup 1 > gpurun_out/a_bench8_rehearsal.json 2> gpurun_out/a_bench8.err; echo "bench8 rc=$?"
tail -c 600 gpurun_out/a_bench8.err
timeout -k 10 400 python tools/shard_probe.py 1.0 8 > gpurun_out/a_shard_probe.json 2> gpurun_out/a_shard_probe.err; echo "probe rc=$?"
