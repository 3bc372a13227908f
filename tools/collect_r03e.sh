#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
for n in 4 8; do
step trace$n 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace$n -o t -- python3 tools/shard_curve.py --ranks $n --steps 100 > $O/trace$n.log 2>&1 || echo "trace rc $?"
tail -3 $O/trace$n.log
python - $n <<'PY'
import csv,glob,collections,sys
n=sys.argv[1]
f=glob.glob(f'gpurun_out/r03e/trace{n}/**/*_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
hs=[r for r in rows if 'hsr::' in r['Kernel_Name']]
print(len(hs),"hsr kernels")
# the submit phase is the last third: take the last 240 kernels (80 steps)
seq=hs[-250:-10]
t0=int(seq[0]['Start_Timestamp'])
for r in seq[:24]:
    nm=r['Kernel_Name'].split('(')[0].replace('void hsr::','')[:28]
    print(f"{nm:30s} q{r['Queue_Id']:>3s} start {(int(r['Start_Timestamp'])-t0)/1e3:9.2f} end {(int(r['End_Timestamp'])-t0)/1e3:9.2f} dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.2f} grid {r['Grid_Size']}")
PY
done
echo done
