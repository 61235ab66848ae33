#!/bin/bash
# scratch probe of the working tree on the GPU box
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_ops.py -m gpu -x -q 2>&1 | tail -6
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_train_model.py tests/test_gpu_boundary.py tests/test_gpu_pvt_train.py -m gpu -x -q 2>&1 | tail -6
CAVP_BENCH_PER_LAYER=$O/layers_train_bf16_b.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
grep -E "bilinear|s2 d1" $O/layers_train_bf16_b.txt
{ tools/ab_trees.sh "" 3; } > $O/ab_session.txt 2>&1; cat $O/ab_session.txt
