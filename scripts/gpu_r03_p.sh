#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for i in 1 2; do
timeout 300 python -X faulthandler -m pytest tests/test_gpu_decoder.py -m gpu -q -p no:cacheprovider -x -s > $OUT/r03p_full$i.log 2>&1
echo "full rc=$?"; grep -v "^  File" $OUT/r03p_full$i.log | tail -4 | cut -c1-300
done
timeout 300 python scripts/bench_decoder.py 2>/dev/null
