#!/bin/bash
# Round-2 profile collection on the GPU box (lands under gpurun_out/r02/, copy what is judged into profiles/).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r02
mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench_train_bf16.json 2> $O/bench_train.err
python bench.py --mode eval --steps 20 --warmup 5 > $O/bench_eval_bf16.json 2> /dev/null
CAVP_BENCH_PER_LAYER=$O/layers_train_bf16.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
for m in train eval; do
  rm -rf $O/prof_$m
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_$m -o $m -- \
     python $GRAFT_REPO_ROOT/bench.py --mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32 > $GRAFT_REPO_ROOT/$O/prof_$m.log 2>&1)
done
ls $O/prof_train | head
