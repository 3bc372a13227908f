#!/bin/bash
# The "batch" and "reduce kernels" sections of tools/collect_profiles.sh on their own (refresh after a change to the batch path).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02final; mkdir -p $O
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_batch -o batch -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 2 > $O/batch_under_rocprof.json 2> $O/trace_batch.log
for c in f32 u16; do timeout -k 10 200 python tools/bench_batch.py --tiles 256 --cube $c > $O/batch_$c.json 2>/dev/null; done
timeout -k 10 200 python tools/bench_batch.py --tiles 64 > $O/batch_f32_t64.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_batch/$c -o p -- python3 tools/bench_batch.py --tiles 256 --no-loop --rounds 1 --reps 3 > /dev/null 2> $O/pmc_batch_$c.log || echo fail; done
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_rs -o rs -- python3 tools/dbg/rs_time.py > /dev/null 2> $O/trace_rs.log
echo done
