#!/bin/bash
# quick GPU check: selected tests only.  Usage: bash tools/gpu_quick.sh <tag> <pytest args...>
set -o pipefail
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=$R/gpurun_out/$TAG
rm -rf $O && mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest -m gpu -x -q "$@" > $O/pytest.log 2>&1; rc=$?
tail -40 $O/pytest.log
exit $rc
