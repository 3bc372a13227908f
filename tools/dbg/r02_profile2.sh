#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
echo "== 5 fresh processes, steps 100"; for i in 1 2 3 4 5; do timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline > $O/fresh_$i.json 2>/dev/null; python -c "import json;d=json.load(open('$O/fresh_$i.json'));print('run $i', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['measured_read_peak'])"; done
echo "== driver-style steps 20"; for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-probe 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print(d['ms_per_step'], d['roofline']['kernel_ms'])"; done
echo "== u16 exact / fast"; timeout -k 10 200 python bench.py --cube u16 --steps 100 > $O/bench_u16.json 2>/dev/null; timeout -k 10 200 python bench.py --cube u16 --u16-fast --steps 100 > $O/bench_u16_fast.json 2>/dev/null; for f in bench_u16 bench_u16_fast; do python -c "import json;d=json.load(open('$O/$f.json'));print('$f', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"; done
echo "== u16 trace + pmc"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_u16 -o bench -- python3 bench.py --cube u16 --no-cpu-baseline --no-probe > /dev/null 2> $O/trace_u16.log
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_u16/$c -o p -- python3 bench.py --cube u16 --steps 10 --warmup 2 --k1-launches 0 --no-cpu-baseline --no-probe > /dev/null 2> $O/pmc_u16_$c.log || echo fail; done
echo "== batch pmc"; for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do n=$(echo $c | cut -d' ' -f1); timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_batch/$n -o p -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 1 --reps 3 > /dev/null 2> $O/pmc_batch_$n.log || echo fail; done
echo "== stamps"; timeout -k 5 120 tools/k1_stamps 1024 1024 64 > $O/k1_stamps.log 2>&1; tail -8 $O/k1_stamps.log
echo done
