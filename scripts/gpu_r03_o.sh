#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -q -p no:cacheprovider > $OUT/r03o_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $OUT/r03o_pytest.log | cut -c1-300 | head -40
timeout 300 python scripts/bench_decoder_kernels.py > $OUT/r03o_decoder_kernels.json 2> $OUT/r03o_decoder_kernels.err
python -c "
import json; d=json.load(open('$OUT/r03o_decoder_kernels.json'))
for k,v in d.items(): print(k, v)"
timeout 300 python scripts/bench_decoder.py > $OUT/r03o_decoder.json 2> $OUT/r03o_decoder.err
echo "bench rc=$?"; tail -3 $OUT/r03o_decoder.err | cut -c1-300; cat $OUT/r03o_decoder.json
cd /tmp && NRHIP_BENCH_DECODER_MODES=hip timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03o -o dec -- python $R/scripts/bench_decoder.py > $OUT/r03o_prof.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03o -name '*.db' | head -1) | head -48 > $OUT/r03o_decoder_kernel_trace.txt
cut -c1-150 $OUT/r03o_decoder_kernel_trace.txt
find $OUT -name '*.db' -path "*prof_r03o*" -delete
