#!/bin/bash
# sweep one tuning environment variable over the headline bench: bash tools/sweep_env.sh VAR v1 v2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/sweep; mkdir -p $O; cd $R
VAR=$1; shift
for b in default "$@"; do
  if [ $b = default ]; then unset $VAR; else export $VAR=$b; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --no-secondary --steps 4 --warmup 2 > $O/b_$b.json 2> $O/b_$b.err || { tail -5 $O/b_$b.err; exit 2; }
  python - <<PY
import json
b=json.loads(open('$O/b_$b.json').read().strip().splitlines()[-1])
print('$VAR=$b', round(b['ms_per_step'],2), {k:round(v,2) for k,v in b['stage_ms'].items() if k in ('par_classes','tree_preorder','tree_class_dfs')})
PY
done
