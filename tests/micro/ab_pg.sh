#!/bin/bash
# A/B of probe builds (lib/probe_<name>.so), reading the step time AND the parameter-gradient launch's own duration
# (bench.py's back-to-back timing): ab_pg.sh 0 nomfma ...
cd "$GRAFT_REPO_ROOT" || exit 1
L=gnn-pressure-estimation_amd/lib
mkdir -p gpurun_out
cp $L/libgatres_hip.so /tmp/orig.so
{
for n in "$@"; do
  cp $L/probe_$n.so $L/libgatres_hip.so
  for rep in 1 2; do
    echo "probe_$n rep $rep: $(GATRES_BENCH_TIMING_ONLY=1 timeout 300 python bench.py --no-cpu-baseline $AB_ARGS 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); r=d["roofline"]; print("ms/step %.5f  window %.1f us  param grads %.2f us" % (d["ms_per_step"], r["avg_launch_us"], (r.get("second_kernel") or {}).get("avg_launch_us", 0)))')"
  done
done
} 2>&1 | tee gpurun_out/ab_pg.txt
cp /tmp/orig.so $L/libgatres_hip.so
