#!/bin/bash
# end-of-round: full GPU suite, smoke, bench lines, traces (every step under timeout)
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $OUT/r03f_pytest.log 2>&1
echo "pytest rc=$?"; tail -2 $OUT/r03f_pytest.log | cut -c1-200
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 600 python bench.py > $OUT/r03f_c1.json 2> $OUT/r03f_c1.err; echo "c1 rc=$?"
timeout 300 python bench.py --config c2 --steps 60 --warmup 10 > $OUT/r03f_c2.json 2> $OUT/r03f_c2.err; echo "c2 rc=$?"
timeout 400 python bench.py --config c3 --steps 30 --warmup 5 > $OUT/r03f_c3.json 2> $OUT/r03f_c3.err; echo "c3 rc=$?"
timeout 400 python bench.py --config c3 --steps 30 --warmup 5 --torch-decoder > $OUT/r03f_c3_torch_decoder.json 2> $OUT/r03f_c3_torch_decoder.err; echo "c3t rc=$?"
timeout 400 python bench.py --config c4 --steps 20 --warmup 5 > $OUT/r03f_c4.json 2> $OUT/r03f_c4.err; echo "c4 rc=$?"
python - <<PY
import json
for n in ("c1","c2","c3","c3_torch_decoder","c4"):
    try:
        d=json.loads([l for l in open("$OUT/r03f_%s.json"%n) if l.startswith("{")][-1])
        print(n, d["value"], d["unit"], "ms", d["ms_per_step"], "frac", d.get("roofline",{}).get("frac"), "traffic", d.get("roofline",{}).get("traffic"))
        if n=="c1":
            print("   train", d["train"]["ms_per_iter"], "train_full", d["train_full"]["ms_per_iter"], d["train_full"].get("hot_path_only",{}).get("ms_per_iter"), "cpu", d["cpu_baseline"]["value"])
    except Exception as e: print(n, "ERR", e)
PY
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03f_tf -o t -- python $R/bench.py --config c3 --steps 10 --warmup 3 > $OUT/prof_r03f_tf.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03f_tf -name '*.db' | head -1) | head -90 > $OUT/r03f_train_full_trace.txt
find $OUT -name '*.db' -path "*prof_r03f*" -delete
NRHIP_BENCH_DECODER_MODES=hip timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03f_dec -o dec -- python $R/scripts/bench_decoder.py > $OUT/r03f_dec_prof.log 2>&1
python $R/scripts/prof_summary.py $(find $OUT/prof_r03f_dec -name '*.db' | head -1) | head -48 > $OUT/r03f_decoder_kernel_trace.txt
find $OUT -name '*.db' -path "*prof_r03f*" -delete
grep "by origin" $OUT/r03f_train_full_trace.txt $OUT/r03f_decoder_kernel_trace.txt | cut -c1-250
cd $R; timeout 200 python scripts/bench_decoder_kernels.py > $OUT/r03f_decoder_kernels.json 2>/dev/null; timeout 200 python scripts/bench_decoder.py 2>/dev/null | tee $OUT/r03f_decoder.json
