#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 300 python scripts/grad_density.py > $OUT/r03l_grad_density.json 2> $OUT/r03l_grad_density.err
echo "density rc=$?"; tail -2 $OUT/r03l_grad_density.err | cut -c1-300; cut -c1-3000 $OUT/r03l_grad_density.json
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/r03l_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/r03l_pytest.log | cut -c1-250
