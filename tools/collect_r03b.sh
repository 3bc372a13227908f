#!/bin/bash
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
step tests 1100 python -m pytest tests -m gpu -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -25 $O/tests.log
step stamps 200 tools/k1_stamps_sets 12 16 > $O/stamps_sets.log 2>&1 || echo "stamps rc $?"
cat $O/stamps_sets.log
step shard 300 python tools/shard_curve.py --graph > $O/shard_curve.log 2>&1 || echo "shard rc $?"
tail -12 $O/shard_curve.log
echo done
