#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for i in 1 2 3; do
timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/r03soak_$i.log 2>&1
echo "run $i rc=$?"; tail -1 $OUT/r03soak_$i.log | cut -c1-150
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
