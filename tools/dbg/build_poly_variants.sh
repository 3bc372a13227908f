#!/bin/bash
set -e
cd "$(dirname "$0")/../../hyperspectral_super-resolution_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
for v in "$@"; do
  name=${v%%:*}; defs=${v#*:}
  /opt/rocm/bin/hipcc $FLAGS $defs -c hsr_poly.hip -o /tmp/hsr_poly_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libhsr_$name.so hsr_srf.o hsr_lib.o /tmp/hsr_poly_$name.o hsr_select.o hsr_ridge.o hsr_resample.o hsr_tile.o hsr_ot.o hsr_chol.o
  echo built $name
done
