cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r06
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > $O/pytest_gpu.txt; cat $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
time python bench.py > $O/bench_default.json 2> $O/bench_default.err; cut -c1-600 $O/bench_default.json
