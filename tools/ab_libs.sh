#!/bin/bash
# same-box comparison of several BUILDS of libcavp_hip.so: tools/ab_libs.sh "<bench.py flags>" rounds lib1.so.bin lib2.so.bin ...
cd $GRAFT_REPO_ROOT
F="$1"; R=$2; shift 2
cp cavp_amd/libcavp_hip.so /tmp/lib_keep.so
for i in $(seq $R); do
  for v in "$@"; do
    cp cavp_amd/$v cavp_amd/libcavp_hip.so
    ms=$(python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-f32 $F 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$v [$F] $ms"
  done
done
cp /tmp/lib_keep.so cavp_amd/libcavp_hip.so
