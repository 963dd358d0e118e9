#!/bin/bash
# PMC passes for the fused kernel at a large batch (GPU box).  Separate --pmc runs, kernel-trace only.
# usage: tools/pmc_probe.sh <outdir> [batch]
set -u
OUT=${1:-gpurun_out/pmc}; B=${2:-1048576}
export TMPDIR=/tmp
mkdir -p $OUT
CMD="python tools/large_step.py $B 10"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM" \
         "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" \
         "GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_I8" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $C --output-format csv -d $OUT/pmc$i -o p -- $CMD > $OUT/pmc$i.log 2>&1 || echo "pass $i failed"
done
python tools/pmc_summary.py $OUT
