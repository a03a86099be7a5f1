#!/bin/bash
# A/B of builds of the library (lib/probe_<name>.so: same sources, different -D flags): for each name given, the parity
# tests of the fused path (optional: AB_TESTS=1) and bench.py twice.  Usage: ab_libs.sh 0 t512 ...
cd "$GRAFT_REPO_ROOT" || exit 1
L=gnn-pressure-estimation_amd/lib
mkdir -p gpurun_out
cp $L/libgatres_hip.so /tmp/orig.so
{
for n in "$@"; do
  cp $L/probe_$n.so $L/libgatres_hip.so
  if [ -n "$AB_TESTS" ] && [ "$n" != 0 ]; then
    timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_headline.py -x -q -m gpu 2>&1 | tail -5
  fi
  for rep in 1 2; do
    echo "probe_$n rep $rep: $(timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.readline()); print(d["ms_per_step"], d.get("parity"))')"
  done
done
} 2>&1 | tee gpurun_out/ab_libs.txt
cp /tmp/orig.so $L/libgatres_hip.so
