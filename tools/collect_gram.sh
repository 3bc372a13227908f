#!/bin/bash
# a9 kernels: parity, kernel trace of tools/bench_ridge.py
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
import csv, glob, collections
f = glob.glob("gpurun_out/gram/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
# the last fit of the second configuration: find the last 3 chol_factor launches -> one per configuration's last repetition
idx = [i for i, r in enumerate(rows) if "chol_factor" in r["Kernel_Name"]]
per = len(idx) // 3
for c in range(3):
    i = idx[(c + 1) * per - 1]
    # walk back to the expand kernel that starts this fit
    j = i
    while j > 0 and "expand_f64" not in rows[j]["Kernel_Name"]: j -= 1
    while j > 0 and int(rows[j]["Start_Timestamp"]) - int(rows[j - 1]["End_Timestamp"]) < 20000 and "predict" not in rows[j - 1]["Kernel_Name"] and "chol_solve" not in rows[j-1]["Kernel_Name"]: j -= 1
    k = i
    while k + 1 < len(rows) and "predict" not in rows[k + 1]["Kernel_Name"] and int(rows[k + 1]["Start_Timestamp"]) - int(rows[k]["End_Timestamp"]) < 20000: k += 1
    t0 = int(rows[j]["Start_Timestamp"])
    print(f"-- configuration {c}: kernels of one fit (start offset us, duration us)")
    for r in rows[j:k + 1]:
        print(f"   {(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f}  {r['Kernel_Name'][:90]}")
PY
