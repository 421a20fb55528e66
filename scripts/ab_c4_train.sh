#!/bin/bash
# A/B of one environment switch on the c4 training step, alternating on one box: usage scripts/ab_c4_train.sh VAR val_a val_b [reps]
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do for v in $A $B; do
  env $VAR=$v python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); t=d['train_step']
print('$VAR=$v', 'c4 training step ms', round(t['ms_per_iter'],3), 'training forward kernel ms', round(t['roofline']['kernel_ms'],4))"
done; done
