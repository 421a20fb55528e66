#!/bin/bash
# kernel-trace stats of the two training sections of bench.py -> gpurun_out/<tag>_train_trace.txt, <tag>_train_full_trace.txt
tag=${1:-r02}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_${tag}_tr -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 40 --train-full-steps 0 > $OUT/prof_${tag}_tr.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_${tag}_tr -name '*.db' | head -1) | head -34 > $OUT/${tag}_train_trace.txt
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_${tag}_tf -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 > $OUT/prof_${tag}_tf.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_${tag}_tf -name '*.db' | head -1) | head -44 > $OUT/${tag}_train_full_trace.txt
find $OUT -name '*.db' -path "*prof_${tag}_t*" -delete
cat $OUT/${tag}_train_trace.txt; cat $OUT/${tag}_train_full_trace.txt
