#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
GLUE_ROWS=70 timeout 400 python scripts/c3_glue_profile.py aten:: > $OUT/r03x_glue.txt 2> $OUT/r03x_glue.err
echo rc=$?; tail -3 $OUT/r03x_glue.err | cut -c1-200; cat $OUT/r03x_glue.txt | cut -c1-200
