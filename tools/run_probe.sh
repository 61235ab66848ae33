#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
{ python tools/probes/conditioned_cross.py condition ab_base /tmp/sd_base.pt
  python tools/probes/conditioned_cross.py condition . /tmp/sd_repo.pt
  for w in base repo; do for t in ab_base .; do python tools/probes/conditioned_cross.py compare $t /tmp/sd_$w.pt; done; done; } 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/conditioned_cross.txt
