#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $OUT/r03q_pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/r03q_pytest.log | cut -c1-250
for v in "" "--torch-decoder"; do
timeout 400 python bench.py --config c3 --steps 20 --warmup 5 $v > $OUT/r03q_c3$v.json 2> $OUT/r03q_c3$v.err
python -c "
import json
d=json.loads([l for l in open('$OUT/r03q_c3$v.json') if l.startswith('{')][-1]); tf=d['train_full']; print('c3 $v ms', d['ms_per_step'], 'decoder', tf['rgb_decoder_fwd_bwd_ms'], 'parity', d.get('parity_rel_l2_vs_oracle'))"
done
NRHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --config c3 --steps 6 --warmup 3 --sparse-exchange > $OUT/r03q_rehearsal_n2_sparse.json 2> $OUT/r03q_rehearsal_n2_sparse.err
echo "rehearsal rc=$?"; tail -2 $OUT/r03q_rehearsal_n2_sparse.err | cut -c1-300
python -c "
import json
d=json.loads([l for l in open('$OUT/r03q_rehearsal_n2_sparse.json') if l.startswith('{')][-1]); tf=d['train_full']; print('rehearsal ms', d['ms_per_step'], tf['grad_exchange'], tf['grad_exchange_wire_bytes_per_rank'], tf['grad_exchange_bytes_per_rank'])"
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03q_tf -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 > $OUT/prof_r03q_tf.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03q_tf -name '*.db' | head -1) | head -75 > $OUT/r03q_train_full_trace.txt
cut -c1-150 $OUT/r03q_train_full_trace.txt | head -60
find $OUT -name '*.db' -path "*prof_r03q*" -delete
