export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/mp; mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o mp -- python3 $GRAFT_REPO_ROOT/tools/bench_match_pair.py > $O/mp.log 2>&1 || { tail -5 $O/mp.log; exit 1; }
cd $GRAFT_REPO_ROOT
grep -v amdgpu $O/mp.log | tail -6
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/mp/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:22]:
    print(f"{r['Name'][:80]:82s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Percentage']} %")
PY
