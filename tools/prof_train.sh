cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_train_bf16.json 2> gpurun_out/bench_train.err
python bench.py --mode eval --steps 20 --warmup 5 > gpurun_out/bench_bf16.json 2> gpurun_out/bench_bf16.err
CAVP_BENCH_PER_LAYER=gpurun_out/layers_train_bf16.txt python bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rm -rf gpurun_out/prof_train; rocprofv3 --kernel-trace --stats -d gpurun_out/prof_train -o train -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/prof_train.log 2>&1
ls -R gpurun_out/prof_train | head
cat gpurun_out/bench_train_bf16.json
