#!/bin/bash
# Round-4 measurement collection on the GPU box (lands under gpurun_out/r04/, the judged files are copied into profiles/).
# usage: tools/prof_r04.sh [part]   part: main (default) | configs | pmc
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04
mkdir -p $O
part=${1:-main}
if [ "$part" = main ]; then
  python bench.py --steps 20 --warmup 5 > $O/bench_train_bf16.json 2> $O/bench_train.err
  CAVP_BENCH_PER_LAYER=$O/layers_train_bf16.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
  rm -rf $O/prof_train
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -o train -- \
     python $GRAFT_REPO_ROOT/bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32 > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1)
  f=$(find $O/prof_train -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_train_bf16.csv
  python tools/summarize_rocprof.py $O/prof_train --igemm-json $O/rocprof_igemm_train_bf16.json > $O/kernel_trace_train_bf16.txt 2>&1
  rm -rf $O/prof_train
  python bench.py --mode eval --steps 50 --warmup 10 > $O/bench_bf16.json 2> $O/bench_eval.err
  head -24 $O/kernel_trace_train_bf16.txt
fi
if [ "$part" = configs ]; then
  python bench.py --deterministic --steps 20 --warmup 5 --no-cpu-baseline --no-f32 > $O/bench_train_bf16_deterministic.json 2>/dev/null
  python bench.py --config c1 --steps 20 --warmup 5 --no-cpu-baseline --no-f32 > $O/bench_c1_train_bf16.json 2>/dev/null
  python bench.py --config c4 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-f32 > $O/bench_c4_pvt_train_bf16.json 2>/dev/null
  python bench.py --config c4 --batch 8 --mode eval --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c4_pvt_eval_bf16.json 2>/dev/null
  python bench.py --config c5 --batch 30 --steps 20 --warmup 5 --no-cpu-baseline --no-f32 > $O/bench_c5_clip_train_bf16.json 2>/dev/null
  python bench.py --trainer-loop --steps 20 --warmup 5 --no-cpu-baseline --no-f32 > $O/bench_trainer_loop_graphed_bf16.json 2>/dev/null
  python bench.py --trainer-loop --no-graph --steps 20 --warmup 5 --no-cpu-baseline --no-f32 > $O/bench_trainer_loop_eager_bf16.json 2>/dev/null
  for f in $O/bench_*_bf16*.json; do echo "$f: $(python -c "import json,sys; d=json.load(open('$f')); print(d['ms_per_step'], d['value'])" 2>/dev/null)"; done
fi
if [ "$part" = pmc ]; then
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-f32 --pmc > $O/bench_train_bf16_pmc.json 2> $O/bench_pmc.err
  ls gpurun_out/ | head; python -c "import json; d=json.load(open('$O/bench_train_bf16_pmc.json')); print(json.dumps(d['roofline'].get('traffic'))); print(json.dumps(d['roofline'].get('step')))"
fi
