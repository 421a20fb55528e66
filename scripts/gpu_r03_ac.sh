#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_actors.py tests/test_gpu_model_glue.py -m gpu -q -p no:cacheprovider -x > $OUT/r03ac_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $OUT/r03ac_pytest.log | cut -c1-300 | head
timeout 400 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/r03ac_c4.json 2> $OUT/r03ac_c4.err
python -c "
import json
d=json.loads([l for l in open('$OUT/r03ac_c4.json') if l.startswith('{')][-1]); print('c4 eval ms', d['ms_per_step'], 'sampler ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'render ms', d['render_roofline']['kernel_ms'], 'train', d['train_step']['ms_per_iter'])"
timeout 200 python scripts/bench_actors.py 100 2>/dev/null | tail -4 | cut -c1-200
