# A/B of float32 K1 wave priorities on one box: old library (HEAD) vs s_setprio levels
set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/u16ab
run() { # label, env...
  local label=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 100 --no-cpu-baseline --no-probe $EXTRA 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('| $label | $EXTRA |', d['ms_per_step'], '|', r['kernel_ms'], '|', r['frac'], '|', r.get('frac_launch_bytes'), '|')
" | tee -a gpurun_out/u16ab/f32.md
}
for EXTRA in "" "--pipeline off"; do
  run old HSR_LIBRARY=$PWD/tools/dbg/libhsr_old.so
  for pr in 0000 1111 3333 3321; do
    run "prio $pr" HSR_DBG_PRIO=$pr
  done
  for pr in 0000 1111; do for pi in 1 3; do
    run "prio $pr issue $pi" HSR_DBG_PRIO=$pr HSR_DBG_PRIO_ISSUE=$pi
  done; done
  run old HSR_LIBRARY=$PWD/tools/dbg/libhsr_old.so
done
