#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train_fused.py tests/test_gpu_model_glue.py -m gpu -q -p no:cacheprovider -x > $OUT/r03aa_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $OUT/r03aa_pytest.log | cut -c1-300 | head
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03aa -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 --no-rgb-decoder > $OUT/prof_r03aa.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03aa -name '*.db' | head -1) | grep -i "power_sampler\|pdf_sample\|interlevel" | cut -c1-120
find $OUT -name '*.db' -path "*prof_r03aa*" -delete
grep "^{" $OUT/prof_r03aa.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('c3 (no decoder, under rocprof) ms', d['ms_per_step'])"
