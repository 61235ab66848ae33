#!/bin/bash
# Same-box A/B of bench.py flag sets (alternating, 3 rounds): bash tools/ab_flags.sh "<flags A>" "<flags B>" [common flags...]
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; shift 2
for r in 1 2 3; do
  for v in "$A" "$B"; do
    t=$(python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 $v "$@" 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "[$v] $t"
  done
done
