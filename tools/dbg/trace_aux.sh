#!/bin/bash
# per-kernel times of tools/bench_aux.py (percentile passes, resamplers, mask, K3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/aux -o aux -- python3 tools/bench_aux.py > gpurun_out/aux.log 2>&1
python3 - <<'PY'
import glob, csv
f = glob.glob("gpurun_out/aux/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "hsr::" in r["Name"]:
        print(f"{float(r['AverageNs'])/1e3:9.1f} us avg  calls {r['Calls']:>5}  {r['Name'][:90]}")
PY
