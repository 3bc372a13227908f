#!/bin/bash
# r03 pass A: full GPU test suite, K1 A/B (LDS-staged fit targets vs r02's inline-asm loads), workgroup timelines per set,
# remaining CU-side PMC passes of the placement map.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
step tests 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "tests rc $?"; tail -15 $O/tests.log
echo "== A/B"; EXTRA="--cold-steps 0" ROUNDS=3 bash tools/dbg/ab_bench.sh prod r02srf > $O/ab.log 2>&1; cat $O/ab.log
step stamps 200 tools/k1_stamps_sets 14 16 > $O/stamps_sets.log 2>&1 || echo "stamps rc $?"
cat $O/stamps_sets.log
n=0
for c in \
  "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum TCP_TCR_RDRET_STALL_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_IB_STALL_sum TCC_NORMAL_EVICT_sum" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" ; do
  n=$((n+1)); D=$O/pmc_$n; mkdir -p $D
  echo "$c" > $D/counters.txt
  step "pmc $n: $c" 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o p -- python3 tools/placement_lab.py map --sets 12 --pitch 16 --log $D/map.json > $D/run.log 2>&1 || echo "pmc $n rc $?"
  python tools/placement_lab.py join $D > $D/join.txt 2>&1 || echo "join $n failed"
  tail -7 $D/join.txt
done
echo done
