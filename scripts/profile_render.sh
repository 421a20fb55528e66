#!/bin/bash
# One gpurun call: kernel-trace stats + separate PMC passes of the headline bench (forward path only), summarised as
# text under gpurun_out/ (copy what should be judged into profiles/).  usage: scripts/profile_render.sh <tag> [bench args]
tag=${1:-r02}; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
BENCH="python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-variants $*"
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_${tag}_trace -o trace -- $BENCH > $OUT/prof_${tag}_trace.log 2>&1
pass() { name=$1; shift; timeout 200 rocprofv3 --pmc "$@" --kernel-trace -d $OUT/prof_${tag}_$name -o $name -- $BENCH > $OUT/prof_${tag}_$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum
pass tcp TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum
pass mfma SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES
pass sq SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY
{
  echo "# $tag: rocprofv3 of: $BENCH"
  python $R/scripts/prof_summary.py $(find $OUT/prof_${tag}_trace -name '*.db' | head -1)
  for p in fetch write tcc tcp mfma sq; do
    echo "== pmc pass $p (render_kernel, mean per dispatch)"
    python $R/scripts/pmc_report.py render_kernel $(find $OUT/prof_${tag}_$p -name '*.db' | head -1)
  done
} > $OUT/${tag}_render_profile.txt 2>&1
find $OUT -name '*.db' -path "*prof_${tag}_*" -delete
tail -60 $OUT/${tag}_render_profile.txt
