#!/bin/bash
# evidence.sh TAG [bench args]: on the GPU box, collect what DESIGN.md / bench.py cite for the headline
# kernel into gpurun_out/evidence_TAG/: bench JSON, rocprofv3 kernel stats, and separate --pmc passes
# (FETCH_SIZE, WRITE_SIZE, two SQ sets).  tools/evidence_summ.py turns them into profiles/ files.
R="$(cd "$(dirname "$0")/.." && pwd)"
TAG=$1; shift
OUT=$R/gpurun_out/evidence_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py "$@" > $OUT/bench.json 2> $OUT/bench.log || { echo bench failed; tail -5 $OUT/bench.log; exit 1; }
cat $OUT/bench.json
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $R/bench.py --steps 12 --warmup 8 --no-cpu-baseline "$@" > $OUT/stats.log 2>&1 || echo FAIL stats
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT" "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES"; do
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$i -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $OUT/pmc_$i.log 2>&1 || echo FAIL pmc $i
  i=$((i+1))
done
echo evidence in $OUT
