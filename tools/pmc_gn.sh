#!/bin/bash
# PMC counters of k_track_gn on the roofline probe (separate passes, counters only: no trace domains).
#   bash tools/pmc_gn.sh <sigma> <ppt> <group> [batch]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
SIG=${1:-0.5}; PPT=${2:-0}; GRP=${3:-0}; B=${4:-64}
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmc_$$
ARGS="--batch $B --level 3 --launches 4 --sigma $SIG --ppt $PPT --group $GRP $PROBE_EXTRA"
i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
  "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" \
  "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" \
  "TD_TD_BUSY_sum TD_TC_STALL_sum SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_TRANS_F32" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-include-regex k_track_gn --output-format csv -d $OUT/p$i -- python3 $R/tools/probe_gn.py $ARGS > $OUT.log 2>&1 || { echo "pass $i failed: $SET"; tail -3 $OUT.log; }
done
PMC_LAST=4 python3 $R/tools/pmc_summary.py "$OUT/p*/*/*counter_collection.csv" > /tmp/pmc_all.txt; cat /tmp/pmc_all.txt
