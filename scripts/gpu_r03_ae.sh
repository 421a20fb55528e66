#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 300 python scripts/decoder_shape_sweep.py 2>&1 | tail -9 | cut -c1-200
