#!/bin/bash
# usage: tools/pmc_conv.sh <shape-substring> <variants>   (GPU box; writes gpurun_out/pmc_conv_*.txt)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
SH=${1:-head0}; V=${2:-1,101,201}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$i -o p -- python tools/bench_conv.py --variants $V --shapes $SH --reps 3 > /tmp/pmc_$i.log 2>&1
  python tools/summarize_rocprof.py /tmp/pmc_$i > gpurun_out/pmc_conv_$i.txt 2>&1
done
cat gpurun_out/pmc_conv_*.txt | grep -v "at::\|elementwise\|^$" | head -150
