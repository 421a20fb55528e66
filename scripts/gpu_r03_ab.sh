#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 300 python -m pytest tests/test_gpu_decoder.py -m gpu -q -p no:cacheprovider -x > $OUT/r03ab_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $OUT/r03ab_pytest.log | cut -c1-300 | head
