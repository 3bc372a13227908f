#!/bin/bash
# Gram kernel: parity, timing, phase timeline
export TMPDIR=/tmp
OUT=gpurun_out/gram; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gram or ridge" > $OUT/tests.log 2>&1 || { tail -20 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for v in default $@; do
  echo "== $v"
  if [ $v = default ]; then unset HSR_LIBRARY; else export HSR_LIBRARY=$PWD/tools/dbg/libhsr_$v.so; fi
  timeout -k 10 120 python tools/time_gram.py || exit 1
done
export HSR_LIBRARY=$PWD/tools/dbg/libhsr_gstamp.so
timeout -k 10 100 python tools/gram_stamps.py
unset HSR_LIBRARY
timeout -k 10 200 python tools/bench_ridge.py 2>&1 | tail -4
