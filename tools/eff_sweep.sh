#!/bin/bash
# End-to-end sweep of the igemm time model's per-tile efficiencies (CAVP_IGEMM_EFF) on ONE box: train / eval ms per step.
cd $GRAFT_REPO_ROOT
run() { t=$(CAVP_IGEMM_EFF="$1" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"); e=$(CAVP_IGEMM_EFF="$1" python bench.py --mode eval --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])"); echo "eff=$1 train $t eval $e"; }
for r in 1 2; do
run "1.0,0.72,0.60,0.75,0.45,0.25,0.45"
run "1.0,0.80,0.70,0.80,0.45,0.25,0.45"
run "1.0,0.85,0.80,0.85,0.50,0.25,0.50"
run "1.0,0.72,0.75,0.75,0.45,0.25,0.45"
run "1.0,0.90,0.70,0.90,0.45,0.25,0.45"
done
