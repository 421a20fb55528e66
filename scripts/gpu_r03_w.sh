#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_decoder.py tests/test_gpu_model_glue.py -m gpu -q -p no:cacheprovider -x -k "decoder" > $OUT/r03w_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $OUT/r03w_pytest.log | cut -c1-300 | head -20
timeout 300 python scripts/bench_decoder.py 2>/dev/null
timeout 400 python bench.py --config c3 --steps 30 --warmup 5 > $OUT/r03w_c3.json 2> $OUT/r03w_c3.err
python -c "
import json
d=json.loads([l for l in open('$OUT/r03w_c3.json') if l.startswith('{')][-1]); tf=d['train_full']; print('c3 ms', d['ms_per_step'], 'decoder', tf['rgb_decoder_fwd_bwd_ms'])"
