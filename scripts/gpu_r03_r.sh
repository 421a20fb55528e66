#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_model_glue.py -m gpu -q -p no:cacheprovider -x -s -k "decoder" > $OUT/r03r_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "rel-L2|passed|failed" $OUT/r03r_pytest.log | cut -c1-250
timeout 300 python scripts/bench_decoder_kernels.py > $OUT/r03r_decoder_kernels.json 2> $OUT/r03r_decoder_kernels.err
python -c "
import json; d=json.load(open('$OUT/r03r_decoder_kernels.json'))
for k,v in d.items(): print(k, v)"
timeout 300 python scripts/bench_decoder.py 2>/dev/null
bash scripts/profile_decoder.sh r03r 2>&1 | cut -c1-150
