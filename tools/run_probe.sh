#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --maxfail=10 2>&1 | tail -25 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
