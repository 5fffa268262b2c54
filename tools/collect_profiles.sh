#!/bin/bash
# Everything profiles/ holds for one state of the code, in one GPU call:
#   bash tools/collect_profiles.sh <tag>  -> gpurun_out/<tag>_{bench.json,kernel_stats.csv,k_track_gn_pmc.txt,traffic.json,launch_schedule.txt}
# The kernel statistics are restricted to the library's kernels (tools/filter_kernel_stats.py; rocprofv3's --kernel-include-regex does
# not filter the --stats table): the synthetic-frame generator (torch elementwise kernels) otherwise fills it.
TAG=${1:-rXX}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p $R/gpurun_out
cd $R && python3 bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err && echo "bench done" &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --kernel-include-regex "dvo::" --output-format csv -d /tmp/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline --pcie-steps 0 --no-secondary > /tmp/prof_$TAG.json 2> /tmp/prof_$TAG.err &&
python3 $R/tools/filter_kernel_stats.py /tmp/prof_$TAG/*/*kernel_stats.csv $R/gpurun_out/${TAG}_kernel_stats.csv && cp /tmp/prof_$TAG.json $R/gpurun_out/${TAG}_bench_under_rocprof.json && echo "stats done" &&
PROBE_EXTRA=--raw bash $R/tools/pmc_gn.sh 0.1 4 2 256 > /dev/null 2>&1; cp /tmp/pmc_all.txt $R/gpurun_out/${TAG}_k_track_gn_pmc_raw_b256.txt && echo "pmc done" &&
bash $R/tools/pmc_traffic.sh > /dev/null 2>&1; cp $R/gpurun_out/traffic.json $R/gpurun_out/${TAG}_traffic.json && echo "traffic done"
# launch schedule of the default bench (per-position kernel times of one step)
cd /tmp && rocprofv3 --kernel-trace --kernel-include-regex "dvo::" --output-format csv -d /tmp/tr_$TAG -- python3 $R/bench.py --no-cpu-baseline --pcie-steps 0 --no-roofline --no-secondary --steps 8 > /dev/null 2>&1 &&
python3 $R/tools/trace_schedule.py "/tmp/tr_$TAG/*/*kernel_trace.csv" 60 > $R/gpurun_out/${TAG}_launch_schedule.txt && echo "schedule done"
