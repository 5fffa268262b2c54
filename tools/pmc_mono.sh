#!/bin/bash
# PMC counters of the mono mapping kernels (1024 sequences): bash tools/pmc_mono.sh <tag> -> gpurun_out/<tag>_mapping_pmc.txt
TAG=${1:-rXX_mono}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
OUT=/tmp/pmcm_$$ && i=0
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-include-regex "k_depth_update|k_propagate|k_regularize|k_promote|k_pyramid" --output-format csv -d $OUT/p$i -- python3 $R/bench.py --workload syn640-mono --batch 1024 --steps 6 --warmup 2 --no-cpu-baseline --no-roofline --pcie-steps 0 > $OUT.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT.log; }
done
python3 $R/tools/pmc_summary.py "$OUT/p*/*/*counter_collection.csv" > $R/gpurun_out/${TAG}_mapping_pmc.txt && echo "pmc done"
