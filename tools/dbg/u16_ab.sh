# A/B of uint16 K1 wave priorities on one box: old library (HEAD) vs s_setprio levels (dots, issue)
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/u16ab
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --cube u16 --steps 100 --no-cpu-baseline --no-probe $EXTRA 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('| $label | $EXTRA |', d['ms_per_step'], '|', r['kernel_ms'], '|', r['frac'], '|', r.get('frac_launch_bytes'), '|')
" | tee -a gpurun_out/u16ab/ab4.md
}
for rep in 1 2; do
for EXTRA in "--pipeline off" "" "--u16-fast"; do
  run old HSR_LIBRARY=$PWD/tools/dbg/libhsr_old.so
  for c in "1 3" "2 3" "1 2" "2 0" "3 0" "1 0" "2 2" "3 3" "2 1"; do
    set -- $c
    run "dots $1 issue $2" HSR_DBG_PRIO=$1$1$1$1 HSR_DBG_PRIO_ISSUE=$2
  done
done
done
for t in f32 u16; do timeout -k 10 200 python tools/bench_batch.py --tiles 256 --cube $t 2>/dev/null | tail -1 | cut -c1-400; HSR_DBG_PRIO=2222 timeout -k 10 200 python tools/bench_batch.py --tiles 256 --cube $t 2>/dev/null | tail -1 | cut -c1-400; done
