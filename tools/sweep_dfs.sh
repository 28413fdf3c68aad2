#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sweep; mkdir -p $O; cd $R
python -m pytest -m gpu -x -q tests/test_gpu_parity.py -k "golden or random_graphs or config2 or hairpin" > $O/pytest.log 2>&1 || { tail -20 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
for b in default 1024 1536 3072 4096; do
  if [ $b = default ]; then unset POVU_HIP_DFS_BLOCKS; else export POVU_HIP_DFS_BLOCKS=$b; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 2 > $O/b_$b.json 2> $O/b_$b.err || { tail -5 $O/b_$b.err; exit 2; }
  python - <<PY
import json
b=json.loads(open('$O/b_$b.json').read().strip().splitlines()[-1])
print('$b', round(b['ms_per_step'],2), 'class_dfs', round(b['stage_ms']['tree_class_dfs'],2), 'pvst', round(b['stage_ms']['par_pvst'],2))
PY
done
