#!/bin/bash
# Round 4, part A: the headline, the exchange pipeline and the mosaic (outputs under gpurun_out/r04final/).
#   /usr/local/graft/bin/gpurun --timeout 1200 -- 'bash tools/collect_r04_a.sh'     then  bash tools/collect_r04_b.sh  in a second call,
#   then, back in the container:  python tools/make_profiles.py r04 gpurun_out/r04final
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04final; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
step "bench default" 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 300 $O/bench_n1.json
echo "== three fresh processes"; for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline > $O/fresh_$i.json 2>/dev/null; done
echo "== driver style (steps 20)"; for i in 1 2 3; do timeout -k 10 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $O/s20_$i.json 2>/dev/null; done
step "no search, no settle" 200 python bench.py --gpus 1 --steps 20 --warmup 5 --placement-trials 0 --settle-ms 0 --no-cpu-baseline --no-probe > $O/s20_plain.json 2>/dev/null
step "kernel trace" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  step "pmc $c" 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$c -o p -- python3 bench.py --steps 10 --warmup 2 --k1-launches 0 --cold-steps 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_$c.log || echo "pmc $c failed"
done
echo "== exchange pipeline (one-rank RCCL communicator through the C ABI)"
step sweep 600 python tools/bench_sweep.py --out $O/exchange_sweep.md -- "" "--force-exchange" "--force-exchange --fake-collective-us 15" "--force-exchange --fake-collective-us 15 --fake-collective-blocks 8" "--force-exchange --coeff-sync broadcast" "--force-exchange --pipeline on" "--height 128" "--force-exchange --height 128" "--force-exchange --height 128 --fake-collective-us 15" "--force-exchange --height 128 --pipeline on" > $O/exchange_sweep.log 2>&1
step "trace exchange" 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fx -o x -- python3 bench.py --force-exchange --fake-collective-us 15 --steps 30 --warmup 5 --cold-steps 0 --k1-launches 0 --no-cpu-baseline --no-probe > $O/bench_fx_under_rocprof.json 2> $O/trace_fx.log
python tools/exchange_timeline.py $O/trace_fx --last 8 > $O/exchange_timeline.md 2>&1
step launcher 300 python bench.py --gpus 1 --launcher --force-exchange --steps 20 --warmup 5 --no-cpu-baseline --no-probe > $O/bench_launcher.json 2> $O/bench_launcher.err
step force_exchange 200 python bench.py --force-exchange --steps 50 --no-cpu-baseline --no-probe > $O/bench_force_exchange.json 2> $O/bench_force_exchange.err
echo "== mosaic"
step mosaic 300 python bench.py --tiles-per-gpu 8 --steps 10 --warmup 2 > $O/bench_mosaic8.json 2> $O/bench_mosaic8.err
step mosaic_batched 300 python bench.py --tiles-per-gpu 8 --steps 10 --warmup 2 --pipeline off > $O/bench_mosaic8_batched.json 2> $O/bench_mosaic8_batched.err
step mosaic4 300 python bench.py --tiles-per-gpu 4 --steps 10 --warmup 2 > $O/bench_mosaic4.json 2> $O/bench_mosaic4.err
echo "== rehearsals: bench.py starting its own ranks (no launcher around it)"
step gloo4 300 python bench.py --gpus 4 --steps 5 --warmup 2 --backend gloo --same-device --height 256 --width 256 > $O/bench_gloo4.json 2> $O/bench_gloo4.err
step strong2 300 python bench.py --gpus 2 --steps 5 --warmup 2 --backend gloo --same-device --scaling strong > $O/bench_strong2.json 2> $O/bench_strong2.err
timeout -k 10 120 python bench.py --gpus 2 --steps 2 > /dev/null 2> $O/two_ranks_one_gpu.err; echo "rccl on one gpu rc $?" >> $O/two_ranks_one_gpu.err
step shard 400 python tools/shard_curve.py --host-cost > $O/shard_curve.log 2>&1
echo done
