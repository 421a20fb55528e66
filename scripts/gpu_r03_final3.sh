#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03h_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/r03h_pytest.log | cut -c1-200
timeout 600 python bench.py > $OUT/r03h_c1.json 2> $OUT/r03h_c1.err; echo "c1 rc=$?"
timeout 400 python bench.py --config c3 --steps 30 --warmup 5 > $OUT/r03h_c3.json 2> $OUT/r03h_c3.err; echo "c3 rc=$?"
timeout 400 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/r03h_c4.json 2> $OUT/r03h_c4.err; echo "c4 rc=$?"
python - <<PY
import json
for n in ("c1","c3","c4"):
    try:
        d=json.loads([l for l in open("$OUT/r03h_%s.json"%n) if l.startswith("{")][-1])
        r=d.get("roofline",{})
        print(n, d["value"], d["unit"], "ms", d["ms_per_step"], r.get("bound"), "frac", r.get("frac"), "kernel_ms", r.get("kernel_ms"), "traffic", r.get("traffic"))
        if n=="c1":
            print("   train", d["train"]["ms_per_iter"], "train_full", d["train_full"]["ms_per_iter"], "cpu", d["cpu_baseline"]["value"], "torch port here", d.get("reference_torch_cpu_port_here"))
        if n=="c4": print("   render_roofline", d["render_roofline"]["frac"], d["render_roofline"]["traffic"], d["render_roofline"]["algorithmic_bytes_per_launch"])
    except Exception as e: print(n, "ERR", e)
PY
