#!/bin/bash
# round 3: eval-time layout of the coarse levels -- tests, A/B on the headline, PMC passes of the headline kernel
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03e_pytest.log 2>&1
echo "pytest rc=$?"; tail -12 $OUT/r03e_pytest.log | cut -c1-250
for rl in 0 1 0 1; do
  NRHIP_EVAL_RELAYOUT=$rl timeout 200 python bench.py --no-cpu-baseline --no-train --steps 200 --warmup 20 > $OUT/r03e_c1_rl$rl.json 2> $OUT/r03e_c1_rl$rl.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/r03e_c1_rl$rl.json') if l.startswith('{')][-1]); print('relayout=$rl ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], d['variants_not_headline']['fp16_table_kernel_us'])"
done
for rl in 0 1; do
  NRHIP_EVAL_RELAYOUT=$rl timeout 200 python bench.py --config c2 --steps 60 --warmup 10 > $OUT/r03e_c2_rl$rl.json 2> $OUT/r03e_c2_rl$rl.err
  python -c "
import json
d=json.loads([l for l in open('$OUT/r03e_c2_rl$rl.json') if l.startswith('{')][-1]); print('c2 relayout=$rl ms_per_step', d['ms_per_step'], 'render_ms', d['render_kernel_ms'], 'sampler_ms', d['roofline']['kernel_ms'])"
done
bash scripts/profile_render.sh r03e > $OUT/r03e_profile_render.log 2>&1
tail -45 $OUT/r03e_profile_render.log | cut -c1-200
