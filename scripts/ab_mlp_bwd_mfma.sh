#!/bin/bash
# What do the fp32 matrix instructions of the fused MLP backward cost?  (round-5 review item 5: pair products in mlp_chain_bwd_wg.)
# The kernel built with its weight-gradient MFMAs replaced by one FMA each (wgfma), with a quarter of the chain's MFMAs (dgq), and
# both -- operand loads and stores unchanged, results wrong: an UPPER BOUND on what any cheaper product form can return.
# usage (GPU box, after the three scripts/build_variant.sh calls): scripts/ab_mlp_bwd_mfma.sh > gpurun_out/ab_mlp_bwd_mfma.txt
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp
for rep in 1 2; do for v in base wgfma dgq both; do
  if [ $v = base ]; then unset NEURAD_HIP_LIB; else export NEURAD_HIP_LIB=$R/neurad_studio_amd/lib/variants/lib_$v.so; fi
  rm -rf $OUT/prof_abm_$v
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_abm_$v -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 --no-graph --no-cpu-baseline > $OUT/prof_abm_$v.log 2>&1
  echo "== $v (rep $rep)"; python $R/scripts/prof_summary.py $(find $OUT/prof_abm_$v -name '*.db' | head -1) | grep "mlp_chain_bwd_wg" | cut -c1-150
  grep -o '"ms_per_step": [0-9.]*' $OUT/prof_abm_$v.log | head -1
  rm -rf $OUT/prof_abm_$v
done; done
