#!/bin/bash
# a9 fit kernels: parity, timing, phase timeline of the Gram kernel
export TMPDIR=/tmp
OUT=gpurun_out/gram; mkdir -p $OUT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gram or ridge or chol" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
timeout -k 10 120 python tools/time_gram.py || exit 1
HSR_LIBRARY=$PWD/tools/dbg/libhsr_gstamp.so timeout -k 10 100 python tools/gram_stamps.py
timeout -k 10 200 python tools/bench_ridge.py 2>&1 | tail -3
