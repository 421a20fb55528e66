#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (no -x: every failure is wanted), the c3 step, its kernel trace
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03a_pytest.log 2>&1
echo "pytest rc=$?"; tail -40 $OUT/r03a_pytest.log
timeout 300 python bench.py --config c3 --steps 10 --warmup 3 > $OUT/r03a_c3.json 2> $OUT/r03a_c3.err
echo "c3 rc=$?"; tail -3 $OUT/r03a_c3.err; cat $OUT/r03a_c3.json | cut -c1-1500
timeout 300 python bench.py --config c3 --steps 10 --warmup 3 --no-rgb-decoder > $OUT/r03a_c3_nodec.json 2> $OUT/r03a_c3_nodec.err
echo "c3 nodec rc=$?"; tail -3 $OUT/r03a_c3_nodec.err; cat $OUT/r03a_c3_nodec.json | cut -c1-600
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03a_tf -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 --no-rgb-decoder > $OUT/prof_r03a_tf.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03a_tf -name '*.db' | head -1) | head -70 > $OUT/r03a_train_full_trace.txt
find $OUT -name '*.db' -path "*prof_r03a_*" -delete
cat $OUT/r03a_train_full_trace.txt | cut -c1-200
