#!/bin/bash
# pmc_ab.sh OUTDIR "label:lib:pairflag" ... : rocprofv3 --pmc passes (sets below) of bench.py per variant
R="$(cd "$(dirname "$0")/.." && pwd)"
OUT=$R/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU2 SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_BUSY_CU_CYCLES SQ_CYCLES" ${PMC_EXTRA:+"$PMC_EXTRA"})
for spec in "$@"; do
  IFS=: read label lib pair <<< "$spec"
  export PINN_HIP_LIB=$R/pinn_depthestimation_amd/$lib PINN_FUSED_PAIR=$pair
  i=0
  for c in "${SETS[@]}"; do
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${label}_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/${label}_$i.log 2>&1 || echo FAIL $label $i
    i=$((i+1))
  done
done
