#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
./scripts/probes/_l1_probe > $OUT/r03f_l1_probe.txt 2>&1; cat $OUT/r03f_l1_probe.txt
timeout 300 python -m pytest tests/test_gpu_eval_layout.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
bash scripts/profile_render.sh r03f > $OUT/r03f_profile_render.log 2>&1
sed -n '/== pmc pass fetch/,$p' $OUT/r03f_profile_render.log | cut -c1-100
timeout 400 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/r03f_c4.json 2> $OUT/r03f_c4.err
echo "c4 rc=$?"; tail -3 $OUT/r03f_c4.err | cut -c1-300; cut -c1-2500 $OUT/r03f_c4.json
