#!/bin/bash
# Second placement pass: phase stamps per candidate set, CU-side counters, per-channel L2 counters (JSON).
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03place2; mkdir -p $O
step() { local name=$1 secs=$2; shift 2; echo "== $name"; timeout -k 10 $secs "$@"; local rc=$?
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "$name timed out: stopping"; exit 1; fi; return $rc; }
step stamps 200 tools/k1_stamps_sets 14 16 > $O/stamps_sets.log 2>&1 || echo "stamps rc $?"
cat $O/stamps_sets.log
n=0
for c in \
  "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum TA_TA_BUSY_sum" \
  "TD_TC_STALL_sum TD_SPI_STALL_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum" \
  "TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCP_LATENCY_sum TCP_TCR_RDRET_STALL_sum" \
  "TCC_HIT_sum TCC_MISS_sum TCC_IB_STALL_sum TCC_NORMAL_EVICT_sum" \
  "TCP_RFIFO_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_UTCL1_LFIFO_FULL_sum TCP_CLIENT_UTCL1_INFLIGHT_sum" \
  "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_LDS" ; do
  n=$((n+1)); D=$O/pmc_$n; mkdir -p $D
  echo "$c" > $D/counters.txt
  step "pmc $n: $c" 280 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o p -- python3 tools/placement_lab.py map --sets 12 --pitch 16 --log $D/map.json > $D/run.log 2>&1 || echo "pmc $n rc $?"
  python tools/placement_lab.py join $D > $D/join.txt 2>&1 || echo "join $n failed"
  tail -7 $D/join.txt
done
# per-channel view: JSON output keeps the counter dimensions
D=$O/chan; mkdir -p $D
step "per-channel json" 280 rocprofv3 --pmc TCC_REQ TCC_BUSY TCC_TAG_STALL TCC_EA0_RDREQ --kernel-trace --output-format json -d $D -o p -- python3 tools/placement_lab.py map --sets 8 --pitch 24 --reps 2 --settle 40 --log $D/map.json > $D/run.log 2>&1 || echo "chan rc $?"
ls -la $D | head
echo done
