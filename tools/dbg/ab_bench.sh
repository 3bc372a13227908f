#!/bin/bash
# A/B of library variants on ONE box: tools/dbg/ab_bench.sh [bench args --] variant...   (prod = the shipped library)
ARGS="--steps 100 --no-cpu-baseline --no-probe"
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do for v in "$@"; do
  if [ $v = prod ]; then L=$PWD/hyperspectral_super-resolution_amd/lib/libhsr_mi355x.so; else L=$PWD/tools/dbg/libhsr_$v.so; fi
  echo -n "$v "; HSR_LIBRARY=$L timeout -k 10 120 python bench.py $ARGS $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'])"
done; done
