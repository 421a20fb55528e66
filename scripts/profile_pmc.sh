#!/bin/bash
# usage: scripts/profile_pmc.sh <tag> <kernel-substr> "<counters pass 1>" "<counters pass 2>" ...   (bench forward path)
tag=$1; kern=$2; shift; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-train $BENCH_ARGS"
cd /tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $ctrs --kernel-trace -d $OUT/pmc_${tag}_$i -o p -- $BENCH > $OUT/pmc_${tag}_$i.log 2>&1
  db=$(find $OUT/pmc_${tag}_$i -name '*.db' | head -1)
  if [ -n "$db" ]; then python $R/scripts/pmc_report.py $kern $db; else echo "pass $i ($ctrs) failed: $(tail -3 $OUT/pmc_${tag}_$i.log)"; fi
  rm -rf $OUT/pmc_${tag}_$i
done
