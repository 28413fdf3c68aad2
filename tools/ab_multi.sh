#!/bin/bash
# Per-kernel comparison of several builds of libpovu_hip.so on the GPU box: a rocprofv3 kernel trace of the headline bench
# (or $WL) with the in-tree library, then with every variant given; prints the kernels matching $SHOW (regex) per build.
# Usage (through gpurun): [WL=hprc-wg] [SHOW='k_uf_tiles|k_class'] bash tools/ab_multi.sh <tag> <variant.so> ...
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
WL=${WL:-hprc-wg}
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
L=$R/povu_amd/lib/libpovu_hip.so
cp $L $O/base.so
cd /tmp && export TMPDIR=/tmp
A="--workload $WL --no-cpu-baseline --no-secondary --no-latency-leg --steps 2 --warmup 1"
i=0
for V in base "$@"; do
  [ $V = base ] || cp $R/$V $L
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$i -- python3 $R/bench.py $A > $O/bench_$i.json 2> $O/err_$i.log || { cp $O/base.so $L; tail -5 $O/err_$i.log; exit 2; }
  cp $(ls $O/kt_$i/*/*kernel_stats.csv | tail -1) $O/kstats_$i.csv
  rm -rf $O/kt_$i
  echo "== $i $V"
  python3 $R/tools/kdiff.py $O/kstats_0.csv $O/kstats_$i.csv 400 | grep -E "passes|${SHOW:-.}"
  i=$((i+1))
done
cp $O/base.so $L; rm -f $O/base.so
