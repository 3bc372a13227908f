export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/sel; mkdir -p $O
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o aux -- python3 $GRAFT_REPO_ROOT/tools/bench_aux.py > $O/aux.log 2>&1 || { tail -5 $O/aux.log; exit 1; }
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections
f = glob.glob("gpurun_out/sel/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if "select_" in r["Kernel_Name"]]
# print the last 14 select launches (one percentile_limits call of each flavour near the end)
seen = collections.OrderedDict()
for r in sel[-60:]:
    n = r["Kernel_Name"][:60] + " grid " + r["Grid_Size_X"] + "x" + r["Grid_Size_Y"]
    seen.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for n, v in seen.items():
    print(f"{n:90s} n={len(v):3d} " + " ".join(f"{x:.1f}" for x in v[-9:]))
PY
grep -i "percentile" $O/aux.log | head
