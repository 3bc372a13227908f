for a in "--cube u16 --tiles-per-gpu 4" "--fused-fit" "--pipeline on" "--pipeline off --cube u16 --u16-fast" "--force-exchange --cube u16" "--force-exchange --coeff-sync broadcast" "--height 512 --width 512 --deg 2" "--height 300 --width 77 --deg 4 --cube u16" "--placement-trials 0 --settle-ms 0 --cold-steps 0" "--tiles-per-gpu 2 --deg 1"; do
  timeout -k 10 200 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-probe $a 2>/tmp/err.txt | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read())
    print('OK  ', '$a', d['ms_per_step'], d['config'].get('pipeline', '')[:40])
except Exception as e:
    print('FAIL', '$a', repr(e)); print(open('/tmp/err.txt').read()[-600:])
"
done
