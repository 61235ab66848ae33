#!/bin/bash
# Round-6 measurement collection on the GPU box (lands under gpurun_out/r06/; the judged files are copied into profiles/).
# usage: tools/prof_r06.sh [part ...]   parts: main | timeline | configs | pmc | wgrad
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
for part in ${@:-main}; do
if [ "$part" = main ]; then
  python bench.py --steps 20 --warmup 5 > $O/bench_train_bf16.json 2> $O/bench_train.err
  CAVP_BENCH_PER_LAYER=$O/layers_train_bf16.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
  rm -rf $O/prof_train
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_train -o train -- \
     python $GRAFT_REPO_ROOT/bench.py --mode train --steps 16 --warmup 4 --no-cpu-baseline --no-roofline --no-f32 --no-eval-leg --no-side-stream > $GRAFT_REPO_ROOT/$O/prof_train.log 2>&1)
  f=$(find $O/prof_train -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp $f $O/rocprofv3_kernel_stats_train_bf16_whole_process.csv
  # the judged summary: graph replays only (no eager warm-up / capture launches in the averages), ONE stream (--no-side-stream: with the
  # audio encoder on its second stream the durations of co-running kernels include their waiting for CUs, and the per-kernel sums
  # stop being comparable with the HIP-event figures, which are taken on one stream too)
  python tools/summarize_rocprof.py $O/prof_train --replays-only --igemm-json $O/rocprof_igemm_train_bf16.json \
     --stats-csv $O/rocprofv3_kernel_stats_train_bf16.csv > $O/kernel_trace_train_bf16.txt 2>&1
  rm -rf $O/prof_train
  python bench.py --mode eval --steps 50 --warmup 10 > $O/bench_bf16.json 2> $O/bench_eval.err
  CAVP_BENCH_PER_LAYER=$O/layers_eval_bf16.txt python bench.py --mode eval --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  head -30 $O/kernel_trace_train_bf16.txt
  cat $O/bench_train_bf16.json | cut -c1-400
fi
if [ "$part" = timeline ]; then
  # one workgroup's s_memtime budget (profile build) for the review's four shapes, 64x64 tile forced (tile id 1 / 11) and the planner's choice
  L=cavp_amd/libcavp_hip_profile.so
  S="l3_1x1_1024_256@14,l3_3x3_256_256@14,T_l2_1x1_512_128@28,T_head0_3x3_304_256@56,l4_1x1_512_2048@14"
  { echo "# product build, planner's tile"; python tools/bench_conv.py --shapes $S --reps 50;
    echo "# profile build, timeline on (CAVP_IGEMM_DBG=256), planner's tile";
    CAVP_IGEMM_DBG=256 python tools/bench_conv.py --lib $L --shapes $S --reps 50;
    echo "# profile build, no MFMA (2) / no DMA issue (8) / neither (10) / neither + no epilogue (26)";
    for d in 2 8 10 26; do echo "## CAVP_IGEMM_DBG=$((d+256))"; CAVP_IGEMM_DBG=$((d+256)) python tools/bench_conv.py --lib $L --shapes $S --reps 50; done; } > $O/igemm_tile_timeline.txt 2>&1
  cat $O/igemm_tile_timeline.txt
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
if [ "$part" = wgrad ]; then
  python tools/bench_wgrad_group.py > $O/wgrad_group.txt 2>&1; tail -20 $O/wgrad_group.txt
fi
done
