#!/bin/bash
# Round profile collection on the GPU box (everything lands under gpurun_out/, copy what is judged into profiles/):
#   bench JSON lines (train bf16 default, eval bf16, eval f32, config #4), per-layer timing dumps, rocprofv3 kernel traces
#   + stats (CSV) of the train and eval steps.  PMC traffic: tools/pmc_traffic.py (separate passes).
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_train_bf16.json 2> gpurun_out/bench_train.err
python bench.py --mode eval --steps 20 --warmup 5 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
python bench.py --mode eval --dtype f32 --steps 10 --warmup 3 > gpurun_out/bench_f32.json 2> /dev/null
python bench.py --config c4 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c4.json 2> /dev/null
CAVP_BENCH_PER_LAYER=gpurun_out/layers_train_bf16.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
CAVP_BENCH_PER_LAYER=gpurun_out/layers_bf16.txt python bench.py --mode eval --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
CAVP_BENCH_PER_LAYER=gpurun_out/layers_f32.txt python bench.py --mode eval --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
for m in train eval; do
  rm -rf gpurun_out/prof_$m
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$m -o $m -- \
     python $GRAFT_REPO_ROOT/bench.py --mode $m --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32 > $GRAFT_REPO_ROOT/gpurun_out/prof_$m.log 2>&1)
done
rm -rf gpurun_out/prof_eval_f32
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_eval_f32 -o evalf32 -- \
   python $GRAFT_REPO_ROOT/bench.py --mode eval --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32 > /dev/null 2>&1)
ls -R gpurun_out/prof_train | head
