#!/bin/bash
# PMC passes of the decoder microbenchmark (conv7x7 forward R=4 and the weight gradient at 40x96x96).  usage: profile_decoder.sh <tag>
tag=${1:-r03}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
BENCH="python $R/scripts/bench_decoder_kernels.py --profile"
cd /tmp
pass() { name=$1; shift; timeout 120 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/prof_${tag}_$name -o $name -- $BENCH > $OUT/prof_${tag}_$name.log 2>&1; }
pass mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES
pass sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
pass act SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC
pass vmem SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
pass fifo SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_WAIT_INST_LDS
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU
pass tcp TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum

{
  echo "# $tag: PMC passes of: $BENCH"
  for k in "conv7_kernelILi4ELb1" conv7_wgrad_kernel; do
  for p in mfma sq act vmem fifo lds tcp; do
    echo "== $k: pmc pass $p (per dispatch)"
    python $R/scripts/pmc_report.py $k $(find $OUT/prof_${tag}_$p -name '*.db' | head -1)
  done
  done
} > $OUT/${tag}_decoder_pmc.txt 2>&1
find $OUT -name '*.db' -path "*prof_${tag}_*" -delete
cat $OUT/${tag}_decoder_pmc.txt
