#!/bin/bash
# diagnostic variants of the library (never shipped): tools/dbg/libhsr_<name>.so, selected with HSR_LIBRARY=...
#   tools/dbg/build_variants.sh "name:-DMACRO=1[:source]" ...      (source defaults to hsr_srf; e.g. hsr_ridge)
set -e
cd "$(dirname "$0")/../../hyperspectral_super-resolution_amd/csrc"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
ALL="hsr_srf hsr_lib hsr_poly hsr_select hsr_ridge hsr_resample hsr_tile hsr_ot hsr_chol hsr_exec hsr_comm"
for v in "$@"; do
  IFS=: read -r name defs src <<< "$v"
  src=${src:-hsr_srf}
  /opt/rocm/bin/hipcc $FLAGS $defs -c $src.hip -o /tmp/${src}_$name.o
  objs=""
  for o in $ALL; do if [ $o = $src ]; then objs="$objs /tmp/${src}_$name.o"; else objs="$objs $o.o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/dbg/libhsr_$name.so $objs -ldl
  echo built $name
done
