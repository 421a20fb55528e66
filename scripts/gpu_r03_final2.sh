#!/bin/bash
# end-of-round, part 2: evidence for the non-headline lines (every step under timeout)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
K="render_kernel<8, 4, 32, true, true, true"
for c in FETCH_SIZE WRITE_SIZE; do
BENCH_ARGS='--config c4 --steps 5 --warmup 2' bash scripts/pmc_pass.sh r03g_c4_$c $c
python scripts/pmc_report.py "$K" $(find $OUT/pmc_r03g_c4_$c -name '*.db' | head -1) | tee $OUT/r03g_c4_pmc_$c.txt
done
find $OUT -name '*.db' -path "*pmc_r03g*" -delete
cd /tmp
for c in c2 c4; do
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03g_$c -o t -- python $R/bench.py --config $c --steps 20 --warmup 5 > $OUT/prof_r03g_$c.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03g_$c -name '*.db' | head -1) | head -40 > $OUT/r03g_${c}_kernel_trace.txt
find $OUT -name '*.db' -path "*prof_r03g_$c*" -delete
head -8 $OUT/r03g_${c}_kernel_trace.txt | cut -c1-150
done
cd $R
bash scripts/profile_decoder.sh r03g > $OUT/r03g_profile_decoder.log 2>&1
grep -A3 "pmc pass mfma" $OUT/r03g_decoder_pmc.txt | cut -c1-120
