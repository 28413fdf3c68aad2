#!/bin/bash
# the wave walk's own counters (build with EXTRA=-DPOVU_WALK_STATS into build/ws): steps by kind and cycles per kind, on the
# workloads the walk dominates.  Usage (through gpurun): bash tools/walk_stats.sh <tag>
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/$1; rm -rf $O; mkdir -p $O
L=$R/povu_amd/lib/libpovu_hip.so
cp $L $O/base.so; cp $R/build/ws/libpovu_hip.so $L
cd $R
for WL in tangled nest circular; do
  timeout -k 10 300 python bench.py --workload $WL --no-cpu-baseline --no-secondary --no-latency-leg --steps 1 --warmup 1 > $O/$WL.out 2> $O/$WL.err
  echo "== $WL"; grep "^walk" $O/$WL.out | sort | uniq -c | sort -rn | head -6
done
cp $O/base.so $L; rm -f $O/base.so
