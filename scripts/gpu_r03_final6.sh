#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
NRHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 --train-steps 20 --train-full-steps 6 > $OUT/r03k_rehearsal_n2.json 2> $OUT/r03k_rehearsal_n2.err
echo "rehearsal rc=$?"
python -c "
import json
d=json.loads([l for l in open('$OUT/r03k_rehearsal_n2.json') if l.startswith('{')][-1]); print('rehearsal n_gpus', d['n_gpus'], 'ms', d['ms_per_step'], 'train', d['train'].get('ms_per_iter', d['train']), 'train_full', d['train_full'].get('ms_per_iter', d['train_full']), d.get('rehearsal','')[:80])"
