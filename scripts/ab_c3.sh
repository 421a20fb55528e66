#!/bin/bash
# A/B of one library switch on the c3 training step, alternating on one box: usage scripts/ab_c3.sh VAR val_a val_b [reps]
VAR=$1; A=$2; B=$3; N=${4:-2}
for i in $(seq $N); do for v in $A $B; do
  env $VAR=$v python bench.py --config c3 --steps 30 --warmup 5 --no-graph 2>/dev/null | python -c "
import sys, json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); tf=d['train_full']
print('$VAR=$v', 'step ms', round(d['ms_per_step'],3), 'field fwd kernel ms', round(d['roofline']['kernel_ms'],4), 'frac', round(d['roofline']['frac'],3))"
done; done
