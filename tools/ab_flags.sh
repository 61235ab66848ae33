cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for v in "--no-rank1-attn" "" "--wgrad-variant 1"; do
    t=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 $v 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$v] train $t"
  done
done
