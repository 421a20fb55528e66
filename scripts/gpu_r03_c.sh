#!/bin/bash
# round 3, third GPU call: workgroup-level record merge in the binned table gradient -- parity tests, A/B of the c3 step, trace
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_fused.py tests/test_gpu_modules.py tests/test_gpu_actors.py -m gpu -q -p no:cacheprovider > $OUT/r03c_pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $OUT/r03c_pytest.log | cut -c1-300
for dd in 1 0; do
  NRHIP_BIN_DEDUPE=$dd timeout 300 python bench.py --config c3 --steps 20 --warmup 5 --no-rgb-decoder > $OUT/r03c_c3_dd$dd.json 2> $OUT/r03c_c3_dd$dd.err
  echo "c3 dedupe=$dd rc=$?"; tail -2 $OUT/r03c_c3_dd$dd.err; python -c "
import json,sys
d=json.loads([l for l in open('$OUT/r03c_c3_dd$dd.json') if l.startswith('{')][-1]); print('ms_per_step', d['ms_per_step'])"
done
timeout 200 python bench.py --no-cpu-baseline --train-full-steps 0 --steps 50 > $OUT/r03c_c1.json 2> $OUT/r03c_c1.err
python -c "
import json
d=json.loads([l for l in open('$OUT/r03c_c1.json') if l.startswith('{')][-1]); print('c1 train', d['train']['ms_per_iter'], d['train'].get('non_saturating'))"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03c_tf -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 --no-rgb-decoder > $OUT/prof_r03c_tf.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03c_tf -name '*.db' | head -1) | head -60 > $OUT/r03c_train_full_trace.txt
find $OUT -name '*.db' -path "*prof_r03c_*" -delete
cut -c1-170 $OUT/r03c_train_full_trace.txt | head -40
