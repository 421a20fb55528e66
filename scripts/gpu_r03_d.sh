#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python scripts/bench_decoder.py > $OUT/r03d_decoder.txt 2>&1
echo "decoder rc=$?"; grep -v "amdgpu.ids" $OUT/r03d_decoder.txt | cut -c1-220 | tail -50
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "binned or skips" > $OUT/r03d_pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $OUT/r03d_pytest.log
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03d_tr -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --train-steps 40 --train-full-steps 0 > $OUT/prof_r03d_tr.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03d_tr -name '*.db' | head -1) | head -40 > $OUT/r03d_train_trace.txt
find $OUT -name '*.db' -path "*prof_r03d_*" -delete
cut -c1-170 $OUT/r03d_train_trace.txt | head -36
