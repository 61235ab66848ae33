#!/bin/bash
# Round-6 same-box A/B runs.  usage: tools/ab_r06.sh part...   parts: tests | quick | step "<flags A>" "<flags B>" ... | trace <flags> | trees | seeds | cross
# (trees / seeds / cross compare the working tree with ab_base/ = a built, untracked checkout of another commit: `git worktree add /tmp/base <commit>`,
# `python -m cavp_amd.build` there, copy bench.py cavp_amd models loss oracle include tools into ab_base/)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
ms() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 --no-eval-leg "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"; }
part=$1; shift
if [ "$part" = tests ]; then
  timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
fi
if [ "$part" = quick ]; then
  timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_train_model.py tests/test_gpu_boundary.py tests/test_gpu_igemm_big.py -m gpu -x -q "$@" 2>&1 | tail -15 > $O/pytest_quick.txt; cat $O/pytest_quick.txt
fi
if [ "$part" = step ]; then
  # every argument is one flag set ("" = default); three alternating rounds
  { for r in 1 2 3; do for f in "$@"; do printf "%-40s %s\n" "[$f]" "$(ms $f)"; done; done; } > $O/ab_step.txt 2>&1
  cat $O/ab_step.txt
fi
# rocprof of a step (graph replays only): per-kernel table -> gpurun_out/r06/kernel_trace_probe.txt
if [ "$part" = trace ]; then
  rm -rf $O/prof_probe
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_probe -o train -- \
     python $GRAFT_REPO_ROOT/bench.py --mode train --steps 16 --warmup 4 --no-cpu-baseline --no-roofline --no-f32 --no-eval-leg "$@" > $GRAFT_REPO_ROOT/$O/prof_probe.log 2>&1)
  python tools/summarize_rocprof.py $O/prof_probe --replays-only > $O/kernel_trace_probe.txt 2>&1
  rm -rf $O/prof_probe
  head -70 $O/kernel_trace_probe.txt | cut -c1-150
fi
# whole-tree A/B against ab_base/: ms per step, train and eval -> gpurun_out/r06/ab_session.txt
if [ "$part" = trees ]; then
  { tools/ab_trees.sh "" 3; tools/ab_trees.sh "--mode eval" 2; } > $O/ab_session.txt 2>&1; cat $O/ab_session.txt
fi
# how far the worst bf16-vs-f32 gradient cosine of test_bf16_teacher_forced_layer_by_layer moves with the input seed, both trees
if [ "$part" = seeds ]; then
  { python tools/probes/bf16_teacher_seeds.py ab_base 11 12 13 14 15 16; python tools/probes/bf16_teacher_seeds.py . 11 12 13 14 15 16; } 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/bf16_teacher_seeds.txt
fi
# conditioned-weights parity figures as a 2 x 2 cross of (weights conditioned by tree X) x (kernels of tree Y)
if [ "$part" = cross ]; then
  { python tools/probes/conditioned_cross.py condition ab_base /tmp/sd_base.pt
    python tools/probes/conditioned_cross.py condition . /tmp/sd_repo.pt
    for w in base repo; do for t in ab_base .; do python tools/probes/conditioned_cross.py compare $t /tmp/sd_$w.pt; done; done; } 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/conditioned_cross.txt
fi
