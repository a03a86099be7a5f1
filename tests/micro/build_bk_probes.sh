#!/bin/bash
# Probe builds of k_blocked.hip (WRONG results, timing only): lib/probe_bk_nog.so (no sparse stage), lib/probe_bk_nod.so (no dense stage)
set -e
P=$(cd "$(dirname "$0")/../../gnn-pressure-estimation_amd" && pwd)
T=$(mktemp -d)
for v in "nog:-DBK_PROBE_NO_GATHER" "nod:-DBK_PROBE_NO_DENSE" $EXTRA; do
  n=${v%%:*}; f=${v#*:}
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -I$P/../include -I$P/csrc $f -c $P/csrc/k_blocked.hip -o $T/k_blocked_$n.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $P/lib/probe_bk_$n.so $(ls $P/build/*.o | grep -v /k_blocked.o) $T/k_blocked_$n.o
  echo $P/lib/probe_bk_$n.so
done
rm -rf $T
