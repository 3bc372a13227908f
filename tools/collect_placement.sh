#!/bin/bash
# Placement mechanism evidence: allocator kinds in one process, then PMC passes over a placement map.
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/collect_placement.sh'
# back in the container: python tools/placement_lab.py join gpurun_out/r03place/pmc_<n>
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03place; mkdir -p $O
step() {   # a timed-out GPU step ends the script: no further GPU work after a kill
  local name=$1 secs=$2; shift 2
  echo "== $name"
  timeout -k 10 $secs "$@"
  local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi
  return $rc
}
step alloc 400 python tools/placement_lab.py alloc --sets 24 --pitch 6 > $O/alloc.log 2>&1 || echo "alloc rc $?"
tail -12 $O/alloc.log
step alloc_arena 300 python tools/placement_lab.py alloc --sets 16 --pitch 6 --arena --kinds torch,hipmalloc,contig,vmm1 > $O/alloc_arena.log 2>&1 || echo "arena rc $?"
tail -8 $O/alloc_arena.log
n=0
for c in \
  "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_TAG_STALL_sum" \
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" \
  "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" \
  "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" \
  "TCC_BUSY_sum TCC_CYCLE_sum TCC_REQ_sum TCC_LATENCY_FIFO_FULL_sum" \
  "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_LEVEL_VMEM SQ_VMEM_TA_ADDR_FIFO_FULL SQ_BUSY_CYCLES" ; do
  n=$((n+1)); D=$O/pmc_$n; mkdir -p $D
  echo "$c" > $D/counters.txt
  step "pmc $n: $c" 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o p -- python3 tools/placement_lab.py map --sets 12 --pitch 16 --log $D/map.json > $D/run.log 2>&1 || echo "pmc $n rc $?"
  python tools/placement_lab.py join $D > $D/join.txt 2>&1 || echo "join $n failed"
  tail -8 $D/join.txt
  # keep the merge-back small: the raw CSV of 12 x 4 + 300 launches is a few hundred KB
done
echo done
