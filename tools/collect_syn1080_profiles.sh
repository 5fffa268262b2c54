#!/bin/bash
# SYN-1080 (BASELINE configs[3]) evidence: bench JSON, rocprofv3 kernel stats, PMC of the finest-level launch, HBM traffic.
#   bash tools/collect_syn1080_profiles.sh <tag> -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,k_track_gn_pmc.txt,traffic.json}
TAG=${1:-rXX_syn1080}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R && python3 bench.py --workload syn1080 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && echo "bench done" &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --workload syn1080 --no-cpu-baseline --pcie-steps 0 --no-secondary > /tmp/prof_$TAG.json 2> /tmp/prof_$TAG.err &&
grep -E '^"Name"|dvo::|rocclr' /tmp/prof_$TAG/*/*kernel_stats.csv > $R/gpurun_out/${TAG}_kernel_stats.csv && echo "stats done" &&
OUT=/tmp/pmcs_$$ && i=0 &&
for SET in \
  "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS" \
  "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE" \
  "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $SET --kernel-include-regex "k_track_gn" --output-format csv -d $OUT/p$i -- python3 $R/bench.py --workload syn1080 --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --pcie-steps 0 --no-secondary --no-roofline > $OUT.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT.log; }
done
python3 $R/tools/pmc_summary.py "$OUT/p*/*/*counter_collection.csv" > $R/gpurun_out/${TAG}_k_track_gn_pmc.txt && echo "pmc done"
