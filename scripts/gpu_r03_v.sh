#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py > $OUT/r03u_c1.json 2> $OUT/r03u_c1.err; echo "c1 rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$OUT/r03u_c1.json") if l.startswith("{")][-1])
print("c1", d["value"], d["unit"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], "kernel_ms", d["roofline"]["kernel_ms"], "traffic", d["roofline"]["traffic"])
print("   train", d["train"]["ms_per_iter"], d["train"]["non_saturating"]["ms_per_iter"], "train_full", d["train_full"]["ms_per_iter"], d["train_full"].get("hot_path_only",{}).get("ms_per_iter"), "cpu", d["cpu_baseline"]["value"], d.get("reference_torch_cpu",{}).get("value"))
PY
for c in FETCH_SIZE WRITE_SIZE; do
BENCH_ARGS='--config c3 --steps 3 --warmup 1 --no-rgb-decoder' bash scripts/pmc_pass.sh r03u_c3_$c $c
python scripts/pmc_report.py "render_kernel<8, 4, 32" $(find $OUT/pmc_r03u_c3_$c -name '*.db' | head -1) | tee $OUT/r03u_c3_pmc_$c.txt
done
find $OUT -name '*.db' -path "*pmc_r03u*" -delete
NRHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 10 --warmup 3 --train-steps 20 --train-full-steps 6 > $OUT/r03u_rehearsal_n2.json 2> $OUT/r03u_rehearsal_n2.err
echo "rehearsal rc=$?"
python -c "
import json
d=json.loads([l for l in open('$OUT/r03u_rehearsal_n2.json') if l.startswith('{')][-1]); print('rehearsal n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'train', d['train'].get('ms_per_iter', d['train']), 'train_full', d['train_full'].get('ms_per_iter', d['train_full']))"
