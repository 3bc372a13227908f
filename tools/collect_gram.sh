#!/bin/bash
# a9 kernels: parity, one fit's kernels from a kernel trace of tools/bench_ridge.py
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/gram; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "gram or ridge or chol" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -1 $OUT/tests.log
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o ridge -- python3 $GRAFT_REPO_ROOT/tools/bench_ridge.py > $OUT/ridge.log 2>&1 || { tail -5 $OUT/ridge.log; exit 1; }
cd $GRAFT_REPO_ROOT
grep n_fit $OUT/ridge.log
python - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gram/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "chol_factor" in r["Kernel_Name"]]
per = len(idx) // 3
i = idx[2 * per - 1]
j = i
while j > 0 and "ridge_stats_partial" not in rows[j]["Kernel_Name"]: j -= 1
k = i
while k + 1 < len(rows) and "ridge_finish" not in rows[k]["Kernel_Name"]: k += 1
t0 = int(rows[j]["Start_Timestamp"])
print("-- kernels of one fit, 29 127 pixels, 32 targets (start offset us, duration us)")
for r in rows[j:k + 1]:
    print(f"   {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  {r['Kernel_Name'][:80]}")
PY
