#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $OUT/r03z_counters.txt 2>&1
grep -c . $OUT/r03z_counters.txt
grep -o "SQ_[A-Z_0-9]*\|TCP_[A-Z_0-9]*\|TA_[A-Z_0-9]*\|TD_[A-Z_0-9]*" $OUT/r03z_counters.txt | sort -u | tr '\n' ' ' | cut -c1-6000
