#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_eval_layout.py -m gpu -q -p no:cacheprovider -k "neighbouring or eval_table" > $OUT/r03h_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/r03h_pytest.log | cut -c1-250
for rm in 1 0 1 0; do
  if [ $rm = 1 ]; then export NRHIP_PROP_FWD_RAYMAJOR=1; else unset NRHIP_PROP_FWD_RAYMAJOR; fi
  timeout 300 python bench.py --config c3 --steps 20 --warmup 5 --no-rgb-decoder > $OUT/r03h_c3_rm$rm.json 2> $OUT/r03h_c3_rm$rm.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/r03h_c3_rm$rm.json') if l.startswith('{')][-1]); print('ray-major=$rm ms_per_step', d['ms_per_step'])"
done
unset NRHIP_PROP_FWD_RAYMAJOR
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03h_tf -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 --no-rgb-decoder > $OUT/prof_r03h_tf.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03h_tf -name '*.db' | head -1) | head -12 > $OUT/r03h_train_full_trace.txt
find $OUT -name '*.db' -path "*prof_r03h_*" -delete
cut -c1-170 $OUT/r03h_train_full_trace.txt
