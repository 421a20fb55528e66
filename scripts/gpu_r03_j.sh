#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for mode in train_fused train_op; do
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03j_$mode -o t -- python $R/scripts/bench_actors.py 100 16384 $mode > $OUT/prof_r03j_$mode.log 2>&1
tail -1 $OUT/prof_r03j_$mode.log
python $R/scripts/prof_summary.py $(find $OUT/prof_r03j_$mode -name '*.db' | head -1) | head -60 > $OUT/r03j_actors_$mode.txt
cut -c1-175 $OUT/r03j_actors_$mode.txt | head -48
done
find $OUT -name '*.db' -path "*prof_r03j_*" -delete
