#!/bin/bash
# End-of-round profile set (GPU box): bench lines of every configuration + PMC passes.  Lands under gpurun_out/r03/final/.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r03/final
mkdir -p $O
python bench.py --mode eval --steps 20 --warmup 5 > $O/bench_bf16.json 2> /dev/null
python bench.py --mode eval --dtype f32 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_f32.json 2> /dev/null
python bench.py --config c1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c1_train_bf16.json 2> /dev/null
python bench.py --config c1 --mode eval --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c1_bf16.json 2> /dev/null
python bench.py --config c4 --batch 8 --steps 20 --warmup 5 > $O/bench_c4_pvt_train_bf16.json 2> /dev/null
python bench.py --config c4 --batch 8 --mode eval --steps 20 --warmup 5 > $O/bench_c4_pvt_eval_bf16.json 2> /dev/null
python bench.py --config c5 --batch 30 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_c5_clip_train_bf16.json 2> /dev/null
python bench.py --deterministic --steps 20 --warmup 5 --no-cpu-baseline --no-f32 > $O/bench_train_bf16_deterministic.json 2> /dev/null
python bench.py --no-side-stream --steps 20 --warmup 5 --no-cpu-baseline --no-f32 --no-roofline > $O/bench_train_bf16_no_side_stream.json 2> /dev/null
python tools/pmc_traffic.py --mode train --out $O/traffic_train_bf16.json > /dev/null 2> $O/pmc_traffic.err
python tools/pmc_mfma.py --mode train --out $O/mfma_train_bf16.json > /dev/null 2> $O/pmc_mfma.err
python bench.py --steps 20 --warmup 5 > $O/bench_train_bf16.json 2> /dev/null
ls -la $O
