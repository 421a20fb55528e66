#!/bin/bash
# round 3, second GPU call: new tests again, the default bench line, c2, c4, the 2-rank rehearsal of bench.py over gloo
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests/test_gpu_train_fused.py tests/test_gpu_dist_rehearsal.py tests/test_gpu_model_glue.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider > $OUT/r03b_pytest.log 2>&1
echo "pytest rc=$?"; tail -15 $OUT/r03b_pytest.log
timeout 600 python bench.py > $OUT/r03b_c1.json 2> $OUT/r03b_c1.err
echo "c1 rc=$?"; tail -3 $OUT/r03b_c1.err; cut -c1-3000 $OUT/r03b_c1.json
timeout 300 python bench.py --config c2 --steps 40 --warmup 10 > $OUT/r03b_c2.json 2> $OUT/r03b_c2.err
echo "c2 rc=$?"; tail -3 $OUT/r03b_c2.err; cut -c1-1500 $OUT/r03b_c2.json
timeout 400 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/r03b_c4.json 2> $OUT/r03b_c4.err
echo "c4 rc=$?"; tail -5 $OUT/r03b_c4.err; cut -c1-2500 $OUT/r03b_c4.json
NRHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --train-steps 20 --train-full-steps 6 > $OUT/r03b_rehearsal_n2.json 2> $OUT/r03b_rehearsal_n2.err
echo "rehearsal rc=$?"; tail -5 $OUT/r03b_rehearsal_n2.err; cut -c1-2500 $OUT/r03b_rehearsal_n2.json
NRHIP_DIST_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --config c3 --steps 6 --warmup 3 --sharded-adam > $OUT/r03b_rehearsal_n2_sharded.json 2> $OUT/r03b_rehearsal_n2_sharded.err
echo "rehearsal sharded rc=$?"; tail -5 $OUT/r03b_rehearsal_n2_sharded.err; cut -c1-1200 $OUT/r03b_rehearsal_n2_sharded.json
