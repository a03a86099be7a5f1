#!/bin/bash
# Probe build of the library: lib/probe_<name>.so = the product library with ONE source recompiled with extra flags.
# Usage: mk_probe.sh <name> <source.hip> [-DFLAG ...]   (name 0 + no flags = a copy of the product library)
# Needs the product objects: python -c "import __graft_entry__ as g; g.build()" first.  Same sources => same build id.
set -e
P=$(cd "$(dirname "$0")/../../gnn-pressure-estimation_amd" && pwd)
name=$1; src=$2; shift 2 || true
if [ -z "$src" ]; then cp $P/lib/libgatres_hip.so $P/lib/probe_$name.so; echo $P/lib/probe_$name.so; exit 0; fi
T=$(mktemp -d)
base=$(basename $src .hip)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffp-contract=off -Wno-unused-function -I$P/../include -I$P/csrc "$@" -c $P/csrc/$src -o $T/$base.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $P/lib/probe_$name.so $(ls $P/build/*.o | grep -v /$base.o) $T/$base.o
rm -rf $T
echo $P/lib/probe_$name.so
