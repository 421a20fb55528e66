#!/bin/bash
# End-of-round measurements in one lease: GPU test suite, the PMC passes behind roofline.traffic (written to profiles/traffic_*.json
# BEFORE the bench lines read them), bench lines c1-c4, kernel traces, the two-rank gloo rehearsals, the round's A/B re-measurements.
# usage: scripts/final_measure.sh <tag>   (outputs under gpurun_out/<tag>_*)
tag=${1:-r06}
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -3 > $OUT/${tag}_gputests.txt; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 >> $OUT/${tag}_gputests.txt
cat $OUT/${tag}_gputests.txt
cd /tmp
pmc() { name=$1; kern=$2; ctr=$3; shift; shift; shift; timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/pmc_${tag}_$name -o p -- "$@" > $OUT/pmc_${tag}_$name.log 2>&1
  echo "== $name: $ctr ($kern)"; python $R/scripts/pmc_report.py "$kern" $(find $OUT/pmc_${tag}_$name -name '*.db' | head -1); rm -rf $OUT/pmc_${tag}_$name; }
if [ -z "$SKIP_PMC" ]; then
{
C1="python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-variants"
pmc c1_fetch render_kernel FETCH_SIZE $C1
pmc c1_write render_kernel WRITE_SIZE $C1
pmc c1_tcc render_kernel "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" $C1
pmc c1_mfma render_kernel "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES" $C1
C2="python $R/bench.py --config c2 --steps 10 --warmup 2 --no-cpu-baseline"
pmc c2_fetch proposal_sampler_kernel FETCH_SIZE $C2
pmc c2_write proposal_sampler_kernel WRITE_SIZE $C2
C3="python $R/bench.py --config c3 --steps 3 --warmup 1 --no-rgb-decoder"
pmc c3_fetch "render_kernel<8, 4, 32, false, false" FETCH_SIZE $C3
pmc c3_write "render_kernel<8, 4, 32, false, false" WRITE_SIZE $C3
C4="python $R/bench.py --config c4 --steps 5 --warmup 2 --train-steps 0"
pmc c4_fetch "render_kernel<8, 4, 32, true, true, true" FETCH_SIZE $C4
pmc c4_write "render_kernel<8, 4, 32, true, true, true" WRITE_SIZE $C4
pmc c4static_fetch "render_kernel<8, 4, 32, true, true, false" FETCH_SIZE $C4
pmc c4s_fetch "proposal_sampler_kernel" FETCH_SIZE $C4
pmc c4s_write "proposal_sampler_kernel" WRITE_SIZE $C4
C4T="python $R/bench.py --config c4 --steps 2 --warmup 1 --train-steps 36"
pmc c4t_fetch "render_kernel<8, 4, 32, true, false, false, 0, false, true" FETCH_SIZE $C4T
pmc c4t_write "render_kernel<8, 4, 32, true, false, false, 0, false, true" WRITE_SIZE $C4T
} > $OUT/${tag}_pmc_traffic.txt 2>&1
cat $OUT/${tag}_pmc_traffic.txt
python $R/scripts/traffic_from_pmc.py $OUT/${tag}_pmc_traffic.txt $tag; cp $R/profiles/traffic_*.json $OUT/
fi
cd $R
timeout 600 python bench.py > $OUT/bench_${tag}_c1.json 2> $OUT/bench_${tag}_c1.err; echo "c1 rc=$?"
for c in c2 c4; do timeout 600 python bench.py --config $c > $OUT/bench_${tag}_$c.json 2> $OUT/bench_${tag}_$c.err; echo "$c rc=$?"; done
# c3: the hand-assembled step (eager + HIP-graph replay) and the same step as `ns-train neurad-hip` runs it (--via-plugin)
timeout 900 python bench.py --config c3 --via-plugin > $OUT/bench_${tag}_c3.json 2> $OUT/bench_${tag}_c3.err; echo "c3 rc=$?"
# `python bench.py --gpus 2` as a plain command (self-launch), over gloo on this box's one GPU: the labelled rehearsal
NRHIP_DIST_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 10 --warmup 3 --train-steps 20 --train-full-steps 6 > $OUT/${tag}_rehearsal_n2_self_launch_gloo_one_gpu.json 2> $OUT/${tag}_rehearsal_n2_self_launch.err; echo "self-launch rehearsal rc=$?"
NRHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 --train-steps 20 --train-full-steps 6 > $OUT/${tag}_rehearsal_n2_gloo_one_gpu.json 2> $OUT/${tag}_rehearsal_n2.err; echo "rehearsal rc=$?"
NRHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 10 --warmup 3 --train-steps 20 --train-full-steps 6 --wire-bf16 > $OUT/${tag}_rehearsal_n2_wire_bf16_gloo_one_gpu.json 2> $OUT/${tag}_rehearsal_n2_wire_bf16.err; echo "rehearsal wire-bf16 rc=$?"
cd /tmp
prof() { name=$1; shift; timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_${tag}_$name -o t -- "$@" > $OUT/prof_${tag}_$name.log 2>&1
  python $R/scripts/prof_summary.py $(find $OUT/prof_${tag}_$name -name '*.db' | head -1) | head -64 > $OUT/${tag}_${name}_kernel_trace.txt; find $OUT/prof_${tag}_$name -name '*.db' -delete; }
prof headline python $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-variants
prof() { name=$1; shift; timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/prof_${tag}_$name -o t -- "$@" > $OUT/prof_${tag}_$name.log 2>&1
  db=$(find $OUT/prof_${tag}_$name -name '*.db' | head -1)
  python $R/scripts/prof_summary.py $db | head -64 > $OUT/${tag}_${name}_kernel_trace.txt
  if [ -n "$GAPS" ]; then python $R/scripts/gpu_gaps.py $db $GAPS 8 > $OUT/${tag}_${name}_gpu_gaps.txt 2>&1; fi
  find $OUT/prof_${tag}_$name -name '*.db' -delete; }
GAPS=60 prof train_full_c3 python $R/bench.py --config c3 --steps 10 --warmup 3 --no-graph
GAPS=60 prof via_plugin_c3 python $R/bench.py --config c3 --steps 10 --warmup 3 --via-plugin-only
prof c2 python $R/bench.py --config c2 --steps 20 --warmup 5 --no-cpu-baseline
prof c4 python $R/bench.py --config c4 --steps 4 --warmup 1 --train-steps 60
NRHIP_BENCH_DECODER_MODES=hip prof decoder python $R/scripts/bench_decoder.py
cd $R
# the round's A/B re-measurements on this box
{
echo "== MLP products of the headline kernel: fp16 pairs (default, NRHIP_MLP_PAIRS=1) vs the fp32 MFMA (0) vs 3-way bf16 split, alternating"
for v in 1 0 1 0 bf16; do if [ $v = bf16 ]; then export NRHIP_MLP_SPLIT_BF16=1 NRHIP_MLP_PAIRS=0; else unset NRHIP_MLP_SPLIT_BF16; export NRHIP_MLP_PAIRS=$v; fi
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-train --no-variants 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('pairs=$v', 'ms_per_step', round(d['ms_per_step'],4), 'value', '%.4g'%d['value'], 'kernel_ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],4))"; done; unset NRHIP_MLP_SPLIT_BF16 NRHIP_MLP_PAIRS
echo "== c4 eval: render stage in ray_order (default) vs data-loader order; fused sampler's in-box pass dense (default) vs inline"
for e in "NRHIP_C4_ORDER_RAYS=1" "NRHIP_C4_ORDER_RAYS=0" "NRHIP_SAMPLER_ACTOR_INLINE=1"; do env $e timeout 300 python bench.py --config c4 --steps 8 --warmup 2 --train-steps 0 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$e', 'eval ms', d['ms_per_step'], 'sampler ms', d['roofline']['kernel_ms'], 'render ms', d['render_roofline']['kernel_ms'])"; done
} > $OUT/${tag}_ab.txt 2>&1
cat $OUT/${tag}_ab.txt
python scripts/show_bench.py $OUT/bench_${tag}_c1.json $OUT/bench_${tag}_c2.json $OUT/bench_${tag}_c3.json $OUT/bench_${tag}_c4.json 2>/dev/null | head -70
