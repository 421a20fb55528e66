#!/bin/bash
# end-of-round measurements: bench lines, PMC passes behind roofline.traffic, c3 kernel trace, N=2 rehearsal
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py > $OUT/r03u_c1.json 2> $OUT/r03u_c1.err; echo "c1 rc=$?"
timeout 300 python bench.py --config c2 --steps 60 --warmup 10 > $OUT/r03u_c2.json 2> $OUT/r03u_c2.err; echo "c2 rc=$?"
timeout 400 python bench.py --config c3 --steps 30 --warmup 5 > $OUT/r03u_c3.json 2> $OUT/r03u_c3.err; echo "c3 rc=$?"
timeout 400 python bench.py --config c3 --steps 30 --warmup 5 --torch-decoder > $OUT/r03u_c3_torch_decoder.json 2> $OUT/r03u_c3_torch_decoder.err; echo "c3t rc=$?"
timeout 400 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/r03u_c4.json 2> $OUT/r03u_c4.err; echo "c4 rc=$?"
python - <<PY
import json
for n in ("c1","c2","c3","c3_torch_decoder","c4"):
    try:
        d=json.loads([l for l in open("$OUT/r03u_%s.json"%n) if l.startswith("{")][-1])
        print(n, d["metric"][:40], d["value"], d["unit"], "ms", d["ms_per_step"], "frac", d.get("roofline",{}).get("frac"), "traffic", d.get("roofline",{}).get("traffic"))
        if n=="c1":
            print("   train", d["train"]["ms_per_iter"], "train_full", d["train_full"]["ms_per_iter"], d["train_full"].get("hot_path_only",{}).get("ms_per_iter"), "cpu", d["cpu_baseline"]["value"])
    except Exception as e: print(n, "ERR", e)
PY
bash scripts/profile_render.sh r03u > $OUT/r03u_profile_render.log 2>&1
sed -n '/== pmc pass fetch/,/== pmc pass tcp/p' $OUT/r03u_render_profile.txt | cut -c1-100
for c in FETCH_SIZE WRITE_SIZE; do
BENCH_ARGS='--config c3 --steps 3 --warmup 1 --no-rgb-decoder' bash scripts/pmc_pass.sh r03u_c3_$c $c
python scripts/pmc_report.py "render_kernel<8, 4, 32" $(find $OUT/pmc_r03u_c3_$c -name '*.db' | head -1) | tee $OUT/r03u_c3_pmc_$c.txt
done
find $OUT -name '*.db' -path "*pmc_r03u*" -delete
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03u_tf -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 > $OUT/prof_r03u_tf.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03u_tf -name '*.db' | head -1) | head -90 > $OUT/r03u_train_full_trace.txt
find $OUT -name '*.db' -path "*prof_r03u*" -delete
cd $R
NRHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 10 --warmup 3 --train-steps 20 --train-full-steps 6 > $OUT/r03u_rehearsal_n2.json 2> $OUT/r03u_rehearsal_n2.err
echo "rehearsal rc=$?"
python -c "
import json
d=json.loads([l for l in open('$OUT/r03u_rehearsal_n2.json') if l.startswith('{')][-1]); print('rehearsal n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'train', d['train'].get('ms_per_iter', d['train']), 'train_full', d['train_full'].get('ms_per_iter', d['train_full']))"
