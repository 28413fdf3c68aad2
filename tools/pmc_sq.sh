#!/bin/bash
# SQ counters per kernel for the headline workload (run on the GPU box through gpurun): is a kernel waiting on memory or issuing?
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/sq
rm -rf $O && mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--workload hprc-wg --no-cpu-baseline --no-secondary --steps 2 --warmup 1"
i=0
for SET in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_LEVEL_WAVES SQ_CYCLES SQ_INST_LEVEL_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_BRANCH GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --output-format csv -d /tmp/sq$i -- python3 $R/bench.py $A > /dev/null 2> $O/err$i.txt || { tail -5 $O/err$i.txt; exit 3; }
  F=$(ls /tmp/sq$i/*/*counter_collection.csv | tail -1)
  for C in $SET; do python3 $R/tools/pmc_summary.py $F $C 4 > $O/$C.json; done
done
echo ok
