#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
step tests 1100 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -8 $O/tests.log
step trace128 200 rocprofv3 --kernel-trace --output-format csv -d $O/trace128 -o t -- python3 tools/shard_curve.py --ranks 8 --steps 200 > $O/trace128.log 2>&1 || echo "trace rc $?"
python - <<'PY'
import csv,glob,collections
f=glob.glob('gpurun_out/r03c/trace128/**/*_kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(f)),key=lambda r:int(r['Start_Timestamp']))
hs=[r for r in rows if 'hsr::' in r['Kernel_Name']]
# last 300 hsr kernels of the plain step() phase: durations and gaps
seq=hs[600:900]
d=collections.defaultdict(list); gaps=collections.defaultdict(list)
for a,b in zip(seq,seq[1:]):
    n=a['Kernel_Name'].split('(')[0].replace('void hsr::','')[:40]
    d[n].append(int(a['End_Timestamp'])-int(a['Start_Timestamp']))
    gaps[n].append(int(b['Start_Timestamp'])-int(a['End_Timestamp']))
for n in d: print(f"{n:42s} n={len(d[n]):4d} dur {sum(d[n])/len(d[n])/1e3:7.2f} us   gap after {sum(gaps[n])/len(gaps[n])/1e3:7.2f} us")
PY
for c in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" ; do
  n=$(echo $c | cut -d' ' -f1); D=$O/ridge_$n; mkdir -p $D
  step "ridge pmc $n" 250 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o p -- python3 tools/bench_ridge.py > $D/run.log 2>&1 || echo "ridge pmc rc $?"
done
step ridge_trace 250 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ridge_trace -o p -- python3 tools/bench_ridge.py > $O/ridge_trace.log 2>&1 || echo rc $?
tail -4 $O/ridge_trace.log
echo done
