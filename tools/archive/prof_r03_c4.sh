#!/bin/bash
# kernel trace of the config-#4 (PVTv2-B5, 512x512, B = 8) training step -> gpurun_out/r03/kernel_trace_c4_<tag>.txt
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-end}
O=gpurun_out/r03
mkdir -p $O
python tools/probes/copy_neighbours.py $O/prof_c4_$T > $O/copy_neighbours_c4_$T.txt 2>&1
rm -rf $O/prof_c4_$T
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4_$T -o c4 -- \
   python $GRAFT_REPO_ROOT/bench.py --config c4 --batch 8 --mode train --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-f32 > $GRAFT_REPO_ROOT/$O/prof_c4_$T.log 2>&1)
python tools/summarize_rocprof.py $O/prof_c4_$T > $O/kernel_trace_c4_$T.txt 2>&1
python tools/probes/copy_neighbours.py $O/prof_c4_$T > $O/copy_neighbours_c4_$T.txt 2>&1
rm -rf $O/prof_c4_$T
head -60 $O/kernel_trace_c4_$T.txt
