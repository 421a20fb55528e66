#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline > $OUT/r03i_c1_$i.json 2> $OUT/r03i_c1_$i.err; echo "c1 rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$OUT/r03i_c1_$i.json") if l.startswith("{")][-1]); tf=d["train_full"]
print("run $i: ms", d["ms_per_step"], "train", d["train"]["ms_per_iter"], "train_full", tf["ms_per_iter"], "allocs", tf["device_allocations_during_the_timed_steps"], tf["allocator_retries_during_the_timed_steps"], "hot", tf["hot_path_only"]["ms_per_iter"], "dec", tf["rgb_decoder_fwd_bwd_ms"])
PY
done
