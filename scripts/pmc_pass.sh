#!/bin/bash
# usage: [BENCH_ARGS="--config c3 --steps 3 --warmup 1"] scripts/pmc_pass.sh <tag> <counter> [<counter>...]
# one rocprofv3 --pmc pass of a short bench run (kernel-trace only; never together with another trace domain)
tag=$1; shift
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
ARGS=${BENCH_ARGS:---steps 10 --warmup 2 --no-cpu-baseline --no-train --no-variants}
cd /tmp
timeout 300 rocprofv3 --pmc "$@" --kernel-trace -d $R/gpurun_out/pmc_$tag -o $tag -- python $R/bench.py $ARGS > $R/gpurun_out/pmc_$tag.log 2>&1
echo "pass $tag rc=$?"
