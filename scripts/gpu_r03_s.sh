#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for d in 0 1 2 3; do
echo "== NRHIP_CONV_DBG=$d (bit0: no tile load, bit1: no stores)"
NRHIP_CONV_DBG=$d timeout 300 python scripts/bench_decoder_kernels.py 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin)
for k,v in d.items():
    if 'R=4' in k and '96' in k or 'R=1' in k and '32x32' in k: print(k, v)"
done
