#!/bin/bash
# diagnostic variants of the library (never shipped): tools/dbg/libhsr_<name>.so, selected with HSR_LIBRARY=...
set -e
cd "$(dirname "$0")/../../hyperspectral_super-resolution_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c hsr_srf.hip -o /tmp/hsr_srf_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libhsr_$name.so /tmp/hsr_srf_$name.o hsr_lib.o hsr_poly.o hsr_select.o hsr_ridge.o hsr_resample.o hsr_tile.o hsr_ot.o hsr_chol.o hsr_exec.o
  echo built $name
done
