#!/bin/bash
# Per-kernel A/B of two builds of libpovu_hip.so on the GPU box (pass times differ by ~1 ms between two runs of the
# same code on one box; kernel durations under rocprofv3 do not): kernel-trace stats of the headline bench with the
# in-tree library (A), then with the variant $1 (B); prints the kernels whose time per pass differs.
# Usage (through gpurun): [AB_SHOW=<regex>] bash tools/ab_kernels.sh <variant.so> <tag> [<workload>]
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$R/$1; TAG=$2; WL=${3:-hprc-wg}
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
L=$R/povu_amd/lib/libpovu_hip.so
cp $L $O/a.so
cd /tmp && export TMPDIR=/tmp
A="--workload $WL --no-cpu-baseline --no-secondary --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_a -- python3 $R/bench.py $A > $O/a.json 2> $O/a.err || { cp $O/a.so $L; exit 2; }
cp $V $L
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_b -- python3 $R/bench.py $A > $O/b.json 2> $O/b.err; rc=$?
cp $O/a.so $L; rm -f $O/a.so
[ $rc -eq 0 ] || exit 3
python3 - $(ls $O/kt_a/*/*kernel_stats.csv | tail -1) $(ls $O/kt_b/*/*kernel_stats.csv | tail -1) <<'PY'
import csv, re, sys
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        nm = re.sub(r'\(.*', '', r['Name']).replace('povu_hip::', '').replace('void ', '')
        d[nm] = d.get(nm, 0.0) + float(r['TotalDurationNs']) / 1e6 / 5  # 5 passes (steps 3 + warmup 1 + 1)
    return d
a, b = load(sys.argv[1]), load(sys.argv[2])
import os
show = os.environ.get('AB_SHOW', '')  # regex of kernels to list whatever the difference
print(f"{'kernel':44s} {'A ms/pass':>10s} {'B ms/pass':>10s} {'A-B':>8s}")
for k in sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, 0) - b.get(k, 0))):
    da, db = a.get(k, 0.0), b.get(k, 0.0)
    if abs(da - db) >= 0.01 or (show and re.search(show, k)):
        print(f"{k[:44]:44s} {da:10.3f} {db:10.3f} {da - db:8.3f}")
print(f"{'sum of all kernels':44s} {sum(a.values()):10.3f} {sum(b.values()):10.3f} {sum(a.values()) - sum(b.values()):8.3f}")
PY
rm -rf $O/kt_a $O/kt_b
