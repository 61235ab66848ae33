#!/bin/bash
# same-box A/B of bench.py flags: tools/ab_r03.sh "<flags A>" "<flags B>" [rounds] -> alternating runs, ms/step of each
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do
  for v in A B; do
    f="${!v}"
    ms=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 $f 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$v [$f] $ms"
  done
done
