#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
step tests 1100 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "tests rc $?"; tail -12 $O/tests.log
step shard 300 python tools/shard_curve.py --graph > $O/shard_curve.log 2>&1 || echo "shard rc $?"
tail -8 $O/shard_curve.log
echo "== A/B constants staging (1024 rows, then 128 rows)"
EXTRA="--cold-steps 0" ROUNDS=3 bash tools/dbg/ab_bench.sh prod constfirst > $O/ab.log 2>&1; cat $O/ab.log
EXTRA="--cold-steps 0 --height 128 --placement-trials 0 --settle-ms 50 --event-every 1" ROUNDS=3 bash tools/dbg/ab_bench.sh prod constfirst > $O/ab128.log 2>&1; cat $O/ab128.log
echo done
