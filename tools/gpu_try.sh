#!/bin/bash
# Experiment session on the GPU box: a pytest selection, then bench lines under a list of environment settings.
# Usage (through gpurun): bash tools/gpu_try.sh <tag> "<pytest -k expression or ''>" "<bench args>" ["ENV=val ENV2=val" ...]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; KEXPR=$2; BARGS=$3; shift 3
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd $R
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
  tail -2 $O/pytest.log
fi
i=0
for ENVS in "$@" ""; do
  [ -z "$ENVS" ] && [ $i -gt 0 ] && break
  i=$((i+1))
  ( export $ENVS; timeout -k 10 600 python bench.py $BARGS > $O/bench_$i.json 2> $O/bench_$i.err ) || { tail -20 $O/bench_$i.err; exit 2; }
  python3 - "$O/bench_$i.json" "$ENVS" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print(sys.argv[2] or "(default)", "| ms", round(b["ms_per_step"], 3), "| frac", round(b["roofline"]["frac"], 4),
      "|", {k: round(v, 2) for k, v in b.get("stage_ms", {}).items() if v >= 0.05})
for k, v in b.get("secondary", {}).items():
    print("   ", k, round(v["ms_per_step"], 2), {a: round(c, 2) for a, c in v["stage_ms"].items() if c >= 0.3})
PY
done
