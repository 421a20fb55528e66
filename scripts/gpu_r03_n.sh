#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd $R
timeout 600 python -m pytest tests/test_gpu_decoder.py -m gpu -q -p no:cacheprovider > $OUT/r03n_pytest.log 2>&1
echo "pytest rc=$?"; grep -E "^E  |passed|failed|Error" $OUT/r03n_pytest.log | cut -c1-300 | head -40
timeout 300 python scripts/bench_decoder.py > $OUT/r03n_decoder.json 2> $OUT/r03n_decoder.err
echo "bench rc=$?"; tail -3 $OUT/r03n_decoder.err | cut -c1-300; cat $OUT/r03n_decoder.json
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_r03n -o dec -- python $R/scripts/bench_decoder.py > $OUT/r03n_prof.log 2>&1
python - <<'PY'
import csv, glob, os
out = os.environ.get("GRAFT_REPO_ROOT", os.getcwd()) + "/gpurun_out"
fs = glob.glob(out + "/prof_r03n/**/*kernel_stats.csv", recursive=True)
if fs:
    rows = list(csv.DictReader(open(fs[0])))
    rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
    with open(out + "/r03n_decoder_kernel_trace.txt", "w") as f:
        for r in rows[:45]:
            line = f'{int(r["Calls"]):6d} {float(r["TotalDurationNs"])/1e3:10.1f} {float(r["AverageNs"])/1e3:9.2f} {r["Percentage"]:>6}%  {r["Name"][:110]}'
            f.write(line + "\n")
    print(open(out + "/r03n_decoder_kernel_trace.txt").read())
PY
