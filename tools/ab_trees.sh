#!/bin/bash
# same-box A/B of two TREES (the working tree vs ab_base/ = a built checkout of another commit, not tracked), alternated R times:
#   tools/ab_trees.sh "<bench.py flags>" [rounds]   -> ms/step of each
cd $GRAFT_REPO_ROOT
F="$1"; R=${2:-3}
ms() { (cd $1 && python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 --no-eval-leg $F 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])"); }
for i in $(seq $R); do
  echo "base [$F] $(ms ab_base)"
  echo "tree [$F] $(ms .)"
done
