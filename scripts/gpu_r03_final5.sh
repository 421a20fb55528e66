#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python bench.py > $OUT/r03j_c1.json 2> $OUT/r03j_c1.err; echo "c1 rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("$OUT/r03j_c1.json") if l.startswith("{")][-1]); tf=d["train_full"]
print("ms", d["ms_per_step"], d["value"], "frac", d["roofline"]["frac"], "train", d["train"]["ms_per_iter"], "train_full", tf["ms_per_iter"], "allocs", tf["device_allocations_during_the_timed_steps"], "hot", tf["hot_path_only"]["ms_per_iter"], "dec", tf["rgb_decoder_fwd_bwd_ms"], "cpu", d["cpu_baseline"]["value"], d["reference_torch_cpu_port_here"]["value"], d["reference_torch_cpu_port_here"]["cores"])
PY
