cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O; rm -rf $O/prof_tl
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_tl -o t -- python $GRAFT_REPO_ROOT/bench.py --trainer-loop --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32 > /dev/null 2>&1)
python tools/summarize_rocprof.py $O/prof_tl > $O/kernel_trace_trainer_loop.txt 2>&1; python tools/probes/kernel_context.py $O/prof_tl direct_copy 4; rm -rf $O/prof_tl
grep -E "at::native|copyBuffer|fillBuffer|cast_|bilinear|total kernel" $O/kernel_trace_trainer_loop.txt | head -30
