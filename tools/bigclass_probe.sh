#!/bin/bash
# parity tests that walk big classes + the two workloads dominated by the class walk
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "tangled or big or nest or tower or fuzz or random or dense" 2>&1 | tail -2
for WL in tangled nest; do
  timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline --no-secondary --steps 3 --warmup 1 | python -c "import sys,json; b=json.loads(sys.stdin.read()); print('$WL', round(b['ms_per_step'],2), round(b['stage_ms']['tree_class_dfs'],2))"
done
