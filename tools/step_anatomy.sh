#!/bin/bash
# Whole-step anatomy with the -DCAVP_PROFILE build (python -m cavp_amd.build --profile): pieces of the igemm kernels switched off for
# EVERY launch of the training step (results are garbage, timings are what is left).  usage (GPU box): bash tools/step_anatomy.sh
cd $GRAFT_REPO_ROOT
run() { CAVP_IGEMM_DBG=$1 python bench.py --lib cavp_amd/libcavp_hip_profile.so --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-f32 2>/dev/null | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])" 2>/dev/null; }
echo "full step (profile build)               $(run 0)"
echo "igemm: no epilogue                      $(run 16)"
echo "igemm: no MFMA                          $(run 2)"
echo "igemm: no DMA instructions              $(run 8)"
echo "igemm: loads out of range (zero fill)   $(run 1)"
echo "igemm: no DMA, no MFMA                  $(run 10)"
echo "igemm: no DMA, no MFMA, no epilogue     $(run 26)"
echo "igemm: no global stores in the epilogue $(run 32)"
echo "igemm: no LDS staging writes            $(run 64)"
