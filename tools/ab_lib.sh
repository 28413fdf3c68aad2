#!/bin/bash
# A/B of two builds of libpovu_hip.so on the GPU box: bench (and optionally a pytest selection) with the in-tree
# library, then with the variant given as $1 (e.g. build/swz/libpovu_hip.so); the in-tree library is put back at the end.
# Usage (through gpurun): bash tools/ab_lib.sh <variant.so> <tag> "<bench args>" ["<pytest -k expr>"]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; TAG=$2; BARGS=$3; KEXPR=$4
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd $R
L=povu_amd/lib/libpovu_hip.so
cp $L $O/a.so
show() { python3 - "$1" "$2" <<'PY'
import json, sys
b = json.load(open(sys.argv[1]))
print(sys.argv[2], "| ms", round(b["ms_per_step"], 3), "|", {k: round(v, 2) for k, v in b.get("stage_ms", {}).items() if v >= 0.05})
for k, v in b.get("secondary", {}).items():
    print("   ", k, round(v["ms_per_step"], 2))
PY
}
timeout -k 10 500 python bench.py $BARGS > $O/bench_a.json 2> $O/bench_a.err || { tail -20 $O/bench_a.err; exit 2; }
show $O/bench_a.json A
cp $V $L
rc=0
if [ -n "$KEXPR" ]; then
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/pytest_b.log 2>&1 || rc=1
  tail -3 $O/pytest_b.log
fi
if [ $rc -eq 0 ]; then
  timeout -k 10 500 python bench.py $BARGS > $O/bench_b.json 2> $O/bench_b.err || { tail -20 $O/bench_b.err; rc=2; }
  [ $rc -eq 0 ] && show $O/bench_b.json B
fi
cp $O/a.so $L
rm -f $O/a.so
exit $rc
