#!/bin/bash
# Round 6: the split launches' first cross-CU barrier -- A/B on one box, one library (GATRES_WINDOW_SYNC_START=1 keeps the barrier).
out=gpurun_out/r06_sync_start_ab.txt
: > $out
run() {
  echo "== $*" >> $out
  env "$@" python bench.py --steps 200 --warmup 20 --repeats 3 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    l = l.strip()
    if l.startswith('{'):
        d = json.loads(l); print(d['ms_per_step'], d['value'])
" >> $out
}
for rep in 1 2 3; do
run A=1
run GATRES_WINDOW_SYNC_START=1
done
cat $out
