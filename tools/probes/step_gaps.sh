cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/r03; mkdir -p $O; rm -rf $O/prof_gap
(cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_gap -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-roofline --no-f32 > /dev/null 2>&1)
python tools/probes/step_gaps.py $O/prof_gap; rm -rf $O/prof_gap
