#!/bin/bash
# A/B of two builds of libcavp_hip.so ON ONE BOX (box-to-box variance is +-3 %, more than most kernel changes):
#   cavp_amd/lib_A.so.bin, cavp_amd/lib_B.so.bin are alternated 3x; prints train / eval ms per step.
# usage (GPU box): bash tools/ab_bench.sh [extra bench.py args]
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in A B; do
    cp cavp_amd/lib_$v.so.bin cavp_amd/libcavp_hip.so
    t=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    e=$(python bench.py --mode eval --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$v train $t eval $e"
  done
done
