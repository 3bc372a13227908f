#!/bin/bash
# Everything the profiles/r02_* files are made of, in one run on the GPU box (outputs under gpurun_out/r02final/).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/collect_profiles.sh'
# then, back in the container:  python tools/make_profiles.py r02 gpurun_out/r02final
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
echo "== bench default"; timeout -k 10 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json
echo "== six fresh processes"; for i in 1 2 3 4 5 6; do timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline > $O/fresh_$i.json 2>/dev/null; done
echo "== driver style (steps 20)"; for i in 1 2 3; do timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $O/s20_$i.json 2>/dev/null; done
timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --settle-ms 0 --no-cpu-baseline --no-probe > $O/s20_nosettle.json 2>/dev/null
echo "== kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.log
for c in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1); echo "== pmc $n"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$n -o p -- python3 bench.py --steps 10 --warmup 2 --k1-launches 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_$n.log || echo "pmc $n failed"
done
echo "== u16"; timeout -k 10 200 python bench.py --cube u16 --steps 100 > $O/bench_u16.json 2>/dev/null; timeout -k 10 200 python bench.py --cube u16 --u16-fast --steps 100 > $O/bench_u16_fast.json 2>/dev/null
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_u16 -o bench -- python3 bench.py --cube u16 --no-cpu-baseline --no-probe > /dev/null 2> $O/trace_u16.log
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_u16/$c -o p -- python3 bench.py --cube u16 --steps 10 --warmup 2 --k1-launches 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_u16_$c.log || echo fail; done
echo "== mosaic"; timeout -k 10 300 python bench.py --tiles-per-gpu 8 --steps 10 --warmup 2 > $O/bench_mosaic8.json 2> $O/bench_mosaic8.err
echo "== rehearsals"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 4 --steps 5 --warmup 2 --backend gloo --same-device --height 256 --width 256 > $O/bench_gloo4.json 2> $O/bench_gloo4.err
timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --steps 2 > /dev/null 2> $O/two_ranks_one_gpu.err
timeout -k 10 200 python bench.py --force-exchange --steps 50 --no-cpu-baseline --no-probe > $O/bench_force_exchange.json 2> $O/bench_force_exchange.err
echo "== batch"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch -o batch -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 2 > $O/batch_under_rocprof.json 2> $O/trace_batch.log
for c in f32 u16; do timeout -k 10 200 python tools/bench_batch.py --tiles 256 --cube $c > $O/batch_$c.json 2>/dev/null; done
timeout -k 10 200 python tools/bench_batch.py --tiles 64 > $O/batch_f32_t64.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_batch/$c -o p -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 1 --reps 3 > /dev/null 2> $O/pmc_batch_$c.log || echo fail; done
echo "== feed"; timeout -k 10 400 python tools/bench_feed.py > $O/feed.log 2>&1
echo "== stamps"; timeout -k 5 120 tools/k1_stamps 1024 1024 64 > $O/k1_stamps.log 2>&1; timeout -k 5 120 tools/k1_stamps 1024 1024 64 1 > $O/k1_stamps_u16.log 2>&1
echo "== probe modes"; timeout -k 10 120 python tools/dbg/probe_modes.py > $O/probe_modes.log 2>&1
echo "== reduce kernels"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_rs -o rs -- python3 tools/dbg/rs_time.py > /dev/null 2> $O/trace_rs.log
echo "== ramp"; timeout -k 10 200 python tools/dbg/ramp.py > $O/ramp.log 2>&1
echo "== address map"; timeout -k 10 200 python tools/dbg/placement_map.py 28 4 > $O/placement_map.log 2>&1
echo "== placement"; timeout -k 10 200 python tools/placement_probe.py > $O/placement_probe.log 2>&1; timeout -k 10 200 python tools/state_probe.py > $O/state_probe.log 2>&1
echo done
