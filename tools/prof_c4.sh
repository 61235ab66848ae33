cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r06; mkdir -p $O; rm -rf $O/prof_c4
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_c4 -o runc -- python $GRAFT_REPO_ROOT/bench.py --config c4 --batch 8 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --no-f32 > $GRAFT_REPO_ROOT/$O/prof_c4.log 2>&1)
python tools/summarize_rocprof.py $O/prof_c4 > $O/kernel_trace_c4_pvt_train_bf16.txt 2>&1
rm -rf $O/prof_c4
grep -E "sra_|total kernel" $O/kernel_trace_c4_pvt_train_bf16.txt
