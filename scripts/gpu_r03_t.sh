#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/r03t_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/r03t_pytest.log | cut -c1-250
timeout 300 python scripts/bench_decoder_kernels.py > $OUT/r03t_decoder_kernels.json 2>/dev/null
timeout 300 python scripts/bench_decoder.py > $OUT/r03t_decoder.json 2>/dev/null; cat $OUT/r03t_decoder.json
timeout 400 python bench.py --config c3 --steps 20 --warmup 5 > $OUT/r03t_c3.json 2> $OUT/r03t_c3.err
python -c "
import json
d=json.loads([l for l in open('$OUT/r03t_c3.json') if l.startswith('{')][-1]); tf=d['train_full']; print('c3 ms', d['ms_per_step'], 'decoder', tf['rgb_decoder_fwd_bwd_ms'])"
cd /tmp && NRHIP_BENCH_DECODER_MODES=hip timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03t -o dec -- python $R/scripts/bench_decoder.py > $OUT/r03t_prof.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03t -name '*.db' | head -1) | head -40 > $OUT/r03t_decoder_kernel_trace.txt
cut -c1-150 $OUT/r03t_decoder_kernel_trace.txt | head -34
find $OUT -name '*.db' -path "*prof_r03t*" -delete
