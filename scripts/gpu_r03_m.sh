#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -q -p no:cacheprovider -x > $OUT/r03m_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/r03m_pytest.log | cut -c1-250
timeout 300 python scripts/bench_decoder_kernels.py > $OUT/r03m_decoder_kernels.json 2> $OUT/r03m_decoder_kernels.err
echo "bench rc=$?"; tail -3 $OUT/r03m_decoder_kernels.err | cut -c1-300; cat $OUT/r03m_decoder_kernels.json
