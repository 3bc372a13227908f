#!/bin/bash
# Everything the profiles/r03_* files are made of, in one run on the GPU box (outputs under gpurun_out/r03final/).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh'
# then, back in the container:  python tools/make_profiles.py r03 gpurun_out/r03final
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03final; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
step "bench default" 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json
echo "== four fresh processes"; for i in 1 2 3 4; do timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline > $O/fresh_$i.json 2>/dev/null; done
echo "== driver style (steps 20)"; for i in 1 2 3; do timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $O/s20_$i.json 2>/dev/null; done
step "no search, no settle" 200 python bench.py --gpus 1 --steps 20 --warmup 5 --placement-trials 0 --settle-ms 0 --no-cpu-baseline --no-probe > $O/s20_plain.json 2>/dev/null
step "kernel trace" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.log
for c in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  step "pmc $n" 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$n -o p -- python3 bench.py --steps 10 --warmup 2 --k1-launches 0 --cold-steps 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_$n.log || echo "pmc $n failed"
done
echo "== u16"; timeout -k 10 200 python bench.py --cube u16 --steps 100 > $O/bench_u16.json 2>/dev/null; timeout -k 10 200 python bench.py --cube u16 --u16-fast --steps 100 > $O/bench_u16_fast.json 2>/dev/null
timeout -k 10 200 python bench.py --cube u16 --steps 100 --pipeline off > $O/bench_u16_off.json 2>/dev/null
step "trace u16" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_u16 -o bench -- python3 bench.py --cube u16 --no-cpu-baseline --no-probe > /dev/null 2> $O/trace_u16.log
for c in FETCH_SIZE WRITE_SIZE; do step "pmc u16 $c" 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_u16/$c -o p -- python3 bench.py --cube u16 --steps 10 --warmup 2 --k1-launches 0 --cold-steps 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_u16_$c.log || echo fail; done
for c in FETCH_SIZE WRITE_SIZE; do step "pmc u16 plain $c" 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_u16_plain/$c -o p -- python3 bench.py --cube u16 --pipeline off --steps 10 --warmup 2 --k1-launches 0 --cold-steps 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_u16_plain_$c.log || echo fail; done
step mosaic 300 python bench.py --tiles-per-gpu 8 --steps 10 --warmup 2 > $O/bench_mosaic8.json 2> $O/bench_mosaic8.err
echo "== rehearsals: bench.py starting its own ranks (no launcher around it)"
step gloo4 300 python bench.py --gpus 4 --steps 5 --warmup 2 --backend gloo --same-device --height 256 --width 256 > $O/bench_gloo4.json 2> $O/bench_gloo4.err
step strong2 300 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --same-device --scaling strong > $O/bench_strong2.json 2> $O/bench_strong2.err
timeout -k 10 120 python bench.py --gpus 2 --steps 2 > /dev/null 2> $O/two_ranks_one_gpu.err; echo "rccl on one gpu rc $?" >> $O/two_ranks_one_gpu.err
step force_exchange 200 python bench.py --force-exchange --steps 50 --no-cpu-baseline --no-probe > $O/bench_force_exchange.json 2> $O/bench_force_exchange.err
step "trace batch" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch -o batch -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 2 > $O/batch_under_rocprof.json 2> $O/trace_batch.log
for c in f32 u16; do timeout -k 10 200 python tools/bench_batch.py --tiles 256 --cube $c > $O/batch_$c.json 2>/dev/null; done
timeout -k 10 200 python tools/bench_aux.py > $O/aux.log 2>&1
timeout -k 10 200 python tools/bench_match_pair.py > $O/match_pair.log 2>&1
echo "== a9"; timeout -k 10 200 python tools/bench_ridge.py > $O/ridge.log 2>&1
timeout -k 10 200 python tools/time_predict.py > $O/time_predict.log 2>&1
step "trace ridge" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_ridge -o p -- python3 tools/bench_ridge.py > /dev/null 2> $O/trace_ridge.log
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1)
  step "ridge pmc $n" 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_ridge/$n -o p -- python3 tools/bench_ridge.py > /dev/null 2> $O/pmc_ridge_$n.log || echo "ridge pmc failed"
done
timeout -k 5 60 tools/chol_stamps 288 32 > $O/chol_stamps.log 2>&1
# phase timeline of the Gram kernel (diagnostic library: tools/dbg/build_variants.sh "gstamp:-DHSR_GRAM_STAMPS:hsr_ridge")
if [ -f tools/dbg/libhsr_gstamp.so ]; then HSR_LIBRARY=$PWD/tools/dbg/libhsr_gstamp.so timeout -k 5 100 python tools/gram_stamps.py > $O/gram_stamps.log 2>&1; fi
echo "== stamps"; timeout -k 5 120 tools/k1_stamps 1024 1024 64 > $O/k1_stamps.log 2>&1
step shard 300 python tools/shard_curve.py --graph --host-cost > $O/shard_curve.log 2>&1
echo done
