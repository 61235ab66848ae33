cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "smallcin or layernorm or linear" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_train_ops.py tests/test_gpu_model.py tests/test_gpu_train_model.py -m gpu -x -q 2>&1 | tail -5
timeout 60 tools/microbench/grid_barrier > $O/grid_barrier.txt 2>&1; cat $O/grid_barrier.txt
CAVP_BENCH_PER_LAYER=$O/layers_train_bf16_b.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-f32 > /dev/null 2>&1
CAVP_BENCH_PER_LAYER=$O/layers_eval_bf16_b.txt python bench.py --mode eval --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
grep -E "layernorm|smallcin" $O/layers_train_bf16_b.txt
grep -E "smallcin|x\(32, 1, 3136, 304\)" $O/layers_eval_bf16_b.txt
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 2>/dev/null | cut -c1-200
