#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_modules.py -m gpu -q -p no:cacheprovider -x -k "neurader or full_size" > $OUT/r03aa_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $OUT/r03aa_pytest.log | cut -c1-300 | head
