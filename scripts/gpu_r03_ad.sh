#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_actors.py tests/test_gpu_model_glue.py tests/test_gpu_modules.py -m gpu -q -p no:cacheprovider -x > $OUT/r03ad_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $OUT/r03ad_pytest.log | cut -c1-300 | head
for v in 0 1; do
NRHIP_C4_ORDER_RAYS=$v timeout 400 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/r03ad_c4_$v.json 2> $OUT/r03ad_c4_$v.err
python -c "
import json
d=json.loads([l for l in open('$OUT/r03ad_c4_$v.json') if l.startswith('{')][-1]); print('order=$v c4 eval ms', d['ms_per_step'], 'sampler ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], 'render ms', d['render_roofline']['kernel_ms'])"
done
