set -o pipefail
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/u16a
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "u16 or uint16 or tile or batch or fused or pipeline" > gpurun_out/u16a/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/u16a/tests.log
for rep in 1 2 3; do
for v in "" "--u16-fast" "--pipeline off"; do
  timeout -k 10 200 python bench.py --cube u16 --steps 100 --no-cpu-baseline $v 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('$v', d['ms_per_step'], r['kernel_ms'], r['frac'], r.get('frac_launch_bytes'))
"
done
done
