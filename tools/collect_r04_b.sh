#!/bin/bash
# Round 4, part B: uint16 tiles, batches, auxiliary kernels, the reference driver, a9 (outputs under gpurun_out/r04final/).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
echo "== u16"; timeout -k 10 200 python bench.py --cube u16 --steps 100 > $O/bench_u16.json 2>/dev/null; timeout -k 10 200 python bench.py --cube u16 --u16-fast --steps 100 > $O/bench_u16_fast.json 2>/dev/null
timeout -k 10 200 python bench.py --cube u16 --steps 100 --pipeline off > $O/bench_u16_off.json 2>/dev/null
step "trace u16" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_u16 -o bench -- python3 bench.py --cube u16 --no-cpu-baseline --no-probe > /dev/null 2> $O/trace_u16.log
for c in FETCH_SIZE WRITE_SIZE; do step "pmc u16 $c" 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_u16/$c -o p -- python3 bench.py --cube u16 --steps 10 --warmup 2 --k1-launches 0 --cold-steps 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_u16_$c.log || echo fail; done
step "trace batch" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch -o batch -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 2 > $O/batch_under_rocprof.json 2> $O/trace_batch.log
for c in f32 u16; do timeout -k 10 200 python tools/bench_batch.py --tiles 256 --cube $c > $O/batch_$c.json 2>/dev/null; done
timeout -k 10 200 python tools/bench_aux.py > $O/aux.log 2>&1
timeout -k 10 200 python tools/bench_match_pair.py > $O/match_pair.log 2>&1
timeout -k 10 200 python tools/dbg/sel_small_c.py > $O/sel_small.log 2>&1
timeout -k 10 200 python tools/dbg/k1_nb13.py > $O/nb13.log 2>&1
echo "== a9"; timeout -k 10 200 python tools/bench_ridge.py > $O/ridge.log 2>&1
timeout -k 10 200 python tools/time_predict.py > $O/time_predict.log 2>&1
step "trace ridge" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ridge -o p -- python3 tools/bench_ridge.py > /dev/null 2> $O/trace_ridge.log
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  step "ridge pmc $n" 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_ridge/$n -o p -- python3 tools/bench_ridge.py > /dev/null 2> $O/pmc_ridge_$n.log || echo "ridge pmc failed"
done
timeout -k 5 60 tools/chol_stamps 288 32 > $O/chol_stamps.log 2>&1
# phase timeline of the Gram kernel (diagnostic library, rebuilt for ABI 5: tools/dbg/build_variants.sh "gstamp:-DHSR_GRAM_STAMPS:hsr_ridge")
if [ -f tools/dbg/libhsr_gstamp.so ]; then HSR_LIBRARY=$PWD/tools/dbg/libhsr_gstamp.so timeout -k 5 100 python tools/gram_stamps.py > $O/gram_stamps.log 2>&1; fi
echo done
