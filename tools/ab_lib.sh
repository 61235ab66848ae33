#!/bin/bash
# same-box A/B of two BUILDS of libcavp_hip.so (cavp_amd/lib_A.so.bin, cavp_amd/lib_B.so.bin), alternated R times:
#   tools/ab_lib.sh "<bench.py flags>" [rounds]   -> ms/step of each
cd $GRAFT_REPO_ROOT
F="$1"; R=${2:-3}
cp cavp_amd/libcavp_hip.so /tmp/lib_keep.so
for i in $(seq $R); do
  for v in A B; do
    cp cavp_amd/lib_$v.so.bin cavp_amd/libcavp_hip.so
    ms=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 $F 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "lib_$v [$F] $ms"
  done
done
cp /tmp/lib_keep.so cavp_amd/libcavp_hip.so
