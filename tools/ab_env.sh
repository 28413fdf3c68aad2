#!/bin/bash
# A/B of one environment switch on the GPU box: the bench line of $WL without and with "$1" (NAME=VALUE) set, plus a kernel
# trace of each (per-kernel difference through tools/kdiff.py).  Usage (through gpurun): [WL=hprc-wg] bash tools/ab_env.sh <tag> NAME=VALUE
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; KV=$2
WL=${WL:-hprc-wg}
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--workload $WL --no-cpu-baseline --no-secondary --no-latency-leg"
python3 $R/bench.py $A --steps 10 --warmup 3 > $O/bench_new.json 2> $O/err_new.log || { tail -5 $O/err_new.log; exit 2; }
export $KV
python3 $R/bench.py $A --steps 10 --warmup 3 > $O/bench_old.json 2> $O/err_old.log || { tail -5 $O/err_old.log; exit 2; }
unset ${KV%%=*}
for i in new old; do
  [ $i = old ] && export $KV
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$i -- python3 $R/bench.py $A --steps 2 --warmup 1 > $O/benchp_$i.json 2> $O/errp_$i.log || { tail -5 $O/errp_$i.log; exit 2; }
  cp $(ls $O/kt_$i/*/*kernel_stats.csv | tail -1) $O/kstats_$i.csv
  rm -rf $O/kt_$i
done
python3 - <<PY
import json
for t in ("new", "old"):
    b = json.loads(open("$O/bench_%s.json" % t).read().strip().splitlines()[-1])
    print(t, "ms_per_step", round(b["ms_per_step"], 3), "stage_ms", {k: round(v, 2) for k, v in b["stage_ms"].items() if v > 0.5})
PY
python3 $R/tools/kdiff.py $O/kstats_old.csv $O/kstats_new.csv 60 | head -60
