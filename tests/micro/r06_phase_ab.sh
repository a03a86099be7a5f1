#!/bin/bash
# Round 6: the window kernel's compile-time facts -- A/B on one box (same library, GATRES_WINDOW_PH_MASK selects which facts a
# launch may use, i.e. the instantiation it takes).  Facts: 0x200 symmetric plan + no consumer workgroups, 0x400 rows of at
# most 6 entries (no edge-at-a-time paths), 0x800 part tables present, 0x1000 parts of at most 64 rows.
out=gpurun_out/r06_phase_ab.txt
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
run GATRES_WINDOW_PH_MASK=65535      # all facts                       -> <32, 1024, 0x1f16>
run GATRES_WINDOW_PH_MASK=4095       # without 0x1000                  -> 0x0f16
run GATRES_WINDOW_PH_MASK=2047       # without 0x1000, 0x800           -> 0x0716
run GATRES_WINDOW_PH_MASK=1023       # 0x200 only                      -> 0x0316
run GATRES_WINDOW_PH_MASK=511        # phases + keep only              -> 0x0116
run GATRES_WINDOW_RUNTIME_PHASES=1   # round 5's kernel (run-time phases)
done
cat $out
