#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_actors.py tests/test_gpu_model_glue.py -m gpu -q -p no:cacheprovider > $OUT/r03k_pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $OUT/r03k_pytest.log | cut -c1-250
timeout 300 python scripts/bench_actors.py 100 > $OUT/r03k_bench_actors.txt 2>&1; tail -7 $OUT/r03k_bench_actors.txt | cut -c1-250
timeout 300 python scripts/bench_actors.py 20 > $OUT/r03k_bench_actors20.txt 2>&1; tail -2 $OUT/r03k_bench_actors20.txt | cut -c1-250
timeout 400 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/r03k_c4.json 2> $OUT/r03k_c4.err
python -c "
import json
d=json.loads([l for l in open('$OUT/r03k_c4.json') if l.startswith('{')][-1]); print('c4 eval ms', d['ms_per_step'], 'train ms', d['train_step']['ms_per_iter'])"
