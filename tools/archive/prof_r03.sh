#!/bin/bash
# Round-3 profile collection on the GPU box (lands under gpurun_out/r03/, copy what is judged into profiles/).
# usage: tools/prof_r03.sh [tag]   (tag names the output files, default "end")
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
T=${1:-end}
O=gpurun_out/r03
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_train_bf16_$T.json 2> $O/bench_train_$T.err
CAVP_BENCH_PER_LAYER=$O/layers_train_bf16_$T.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
rm -rf $O/prof_train_$T
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train_$T -o train -- \
   python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32 > $GRAFT_REPO_ROOT/$O/prof_train_$T.log 2>&1)
f=$(find $O/prof_train_$T -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_train_bf16_$T.csv
t=$(find $O/prof_train_$T -name '*kernel_trace.csv' | head -1)
python tools/summarize_rocprof.py $O/prof_train_$T > $O/kernel_trace_train_bf16_$T.txt 2>&1
rm -rf $O/prof_train_$T   # the raw trace is tens of MB
head -30 $O/kernel_trace_train_bf16_$T.txt
