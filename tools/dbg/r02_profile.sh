#!/bin/bash
# round-2 profile collection on the GPU box (outputs under gpurun_out/r02/)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
echo "== tests (match_pair, batch)"; timeout -k 10 300 python -m pytest tests -m gpu -q -k "match_pair or batch" 2>&1 | tail -2
echo "== bench default"; timeout -k 10 300 python bench.py > $O/bench_n1.json 2> $O/bench_n1.err; tail -c 600 $O/bench_n1.json
echo "== bench steps 20 (driver style)"; timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/bench_n1_s20.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_n1_s20.json'));print(d['ms_per_step'], d['roofline']['kernel_ms'])"
echo "== kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o bench -- python3 bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/trace.log; ls $O/trace | head
for c in FETCH_SIZE WRITE_SIZE "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  n=$(echo $c | cut -d' ' -f1); echo "== pmc $n"
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc/$n -o p -- python3 bench.py --steps 10 --warmup 2 --k1-launches 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_$n.log || echo "pmc $n failed"
done
echo "== mosaic 8 tiles"; timeout -k 10 300 python bench.py --tiles-per-gpu 8 --steps 10 --warmup 2 > $O/bench_mosaic8.json 2> $O/bench_mosaic8.err; tail -c 400 $O/bench_mosaic8.json; tail -2 $O/bench_mosaic8.err
echo "== u16"; timeout -k 10 300 python bench.py --cube u16 --steps 50 > $O/bench_u16.json 2>/dev/null; python -c "import json;d=json.load(open('$O/bench_u16.json'));print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
echo "== 4 ranks gloo same device"; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 4 --steps 5 --warmup 2 --backend gloo --same-device --height 256 --width 256 > $O/bench_gloo4.json 2> $O/bench_gloo4.err; tail -c 500 $O/bench_gloo4.json; tail -2 $O/bench_gloo4.err
echo "== too few devices"; timeout -k 10 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29556 bench.py --gpus 2 --steps 2 > /dev/null 2> $O/two_ranks_one_gpu.err; grep -m1 "\[bench\]" $O/two_ranks_one_gpu.err
echo "== batch trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch -o batch -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 2 > $O/batch_under_rocprof.json 2> $O/trace_batch.log
for c in f32 u16; do timeout -k 10 200 python tools/bench_batch.py --tiles 256 --cube $c > $O/batch_$c.json 2>/dev/null; cat $O/batch_$c.json; done
timeout -k 10 200 python tools/bench_batch.py --tiles 64 > $O/batch_f32_t64.json 2>/dev/null; cat $O/batch_f32_t64.json
echo done
