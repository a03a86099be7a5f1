#!/bin/bash
# Probe builds of the library (not product code): lib/probe_st.so = libgatres_hip.so with k_proj.hip compiled -DPROJ_STAMPS
# (per-wave wall-clock stamps inside proj_bf16_stream_kernel, read by tests/micro/proj_probe.py --stamps).
# Needs the product library's objects: python -c "import __graft_entry__ as g; g.build()" first.
set -e
P=$(cd "$(dirname "$0")/../../gnn-pressure-estimation_amd" && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -I$P/../include -I$P/csrc -DPROJ_STAMPS -c $P/csrc/k_proj.hip -o $T/k_proj_st.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $P/lib/probe_st.so $(ls $P/build/*.o | grep -v /k_proj.o) $T/k_proj_st.o
rm -rf $T
echo $P/lib/probe_st.so
