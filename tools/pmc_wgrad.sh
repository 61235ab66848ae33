#!/bin/bash
# usage: tools/pmc_wgrad.sh <shape-substring> [splitk] [counter sets...]  (GPU box; prints per-kernel counter sums)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SH=${1:-ca_fc1}; SK=${2:-0}; shift; shift
i=0
for C in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmcw_$i
  timeout 150 rocprofv3 --pmc $C --output-format csv -d /tmp/pmcw_$i -o p -- python tools/bench_wgrad.py --shapes $SH --splitk $SK --reps 2 > /tmp/pmcw_$i.log 2>&1 || echo "set $i ($C) timed out / failed"
  python tools/summarize_rocprof.py /tmp/pmcw_$i 2>&1 | grep "wgrad_kernel"
done
