#!/bin/bash
# Run on the GPU box (through gpurun): for the headline workload (config 4, `hprc-wg`) and config 2 (`chain`):
# the bench line, a rocprofv3 kernel trace with stats, and the two PMC passes (FETCH_SIZE / WRITE_SIZE in separate
# runs, as MI355X_MICROARCH.md prescribes) of the same command.  Everything lands in gpurun_out/final/<workload>/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for WL in hprc-wg chain; do
  O=$R/gpurun_out/final/$WL
  rm -rf $O && mkdir -p $O
  if [ $WL = hprc-wg ]; then
    python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
  else
    python3 $R/bench.py --workload $WL --no-secondary > $O/bench.json 2> $O/bench.err || exit 1
  fi
  A="--workload $WL --no-cpu-baseline --no-secondary --no-overlap --no-latency-leg --steps 2 --warmup 1"   # 4 decompose passes + one upload
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $A > $O/bench_under_rocprof.json 2> $O/kt.err || exit 2
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py $A > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 3
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py $A > $O/pmc_write.json 2> $O/pmc_write.err || exit 4
  python3 $R/tools/pmc_summary.py $(ls $O/pmc_fetch/*/*counter_collection.csv | tail -1) FETCH_SIZE 4 > $O/fetch_summary.json
  python3 $R/tools/pmc_summary.py $(ls $O/pmc_write/*/*counter_collection.csv | tail -1) WRITE_SIZE 4 > $O/write_summary.json
  python3 $R/tools/trace_timeline.py $(ls $O/kt/*/*kernel_trace.csv | tail -1) x > $O/timeline_summary.txt
  cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/kernel_stats.csv
  # keep the merge small: the per-dispatch tables are tens of MB
  rm -f $O/pmc_fetch/*/*counter_collection.csv $O/pmc_write/*/*counter_collection.csv $O/kt/*/*kernel_trace.csv
  echo "== $WL"; cat $O/timeline_summary.txt; head -c 600 $O/bench.json; echo
done
# the workloads the class walks dominate: a kernel trace of the walk that ships (BASELINE config 5, the tangled graph, the
# circular component)
for WL in nest tangled circular; do
  O=$R/gpurun_out/final/$WL
  rm -rf $O && mkdir -p $O
  A="--workload $WL --no-cpu-baseline --no-secondary --no-overlap --no-latency-leg --steps 2 --warmup 1"
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py $A > $O/bench_under_rocprof.json 2> $O/kt.err || exit 5
  cp $(ls $O/kt/*/*kernel_stats.csv | tail -1) $O/kernel_stats.csv
  rm -rf $O/kt
  echo "== $WL"; head -c 400 $O/bench_under_rocprof.json; echo
done
