#!/bin/bash
# Run on the GPU box (through gpurun): bench line, rocprofv3 kernel stats and the two PMC passes of the
# same command, all into gpurun_out/final/.
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/final
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 $R/bench.py --no-cpu-baseline --steps 10 --warmup 2 > $O/bench_under_rocprof.json 2> $O/kt.err || exit 2
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_fetch.json 2> $O/pmc_fetch.err || exit 3
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $O/pmc_write.json 2> $O/pmc_write.err || exit 4
# bench.py --steps 2 --warmup 1 runs 4 decompose passes (1 warm-up, 2 timed, 1 for the stage breakdown) + one upload
python3 $R/tools/pmc_summary.py $(ls $O/pmc_fetch/*/*counter_collection.csv | tail -1) FETCH_SIZE 4 > $O/fetch_summary.json
python3 $R/tools/pmc_summary.py $(ls $O/pmc_write/*/*counter_collection.csv | tail -1) WRITE_SIZE 4 > $O/write_summary.json
python3 $R/tools/trace_timeline.py $(ls $O/kt/*/*kernel_trace.csv | tail -1) x > $O/timeline_summary.txt
cat $O/bench.json; cat $O/fetch_summary.json; cat $O/write_summary.json; cat $O/timeline_summary.txt
# keep the merge small: the per-dispatch PMC tables are tens of MB
rm -f $O/pmc_fetch/*/*counter_collection.csv $O/pmc_write/*/*counter_collection.csv
