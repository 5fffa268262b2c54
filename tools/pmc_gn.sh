R=$GRAFT_REPO_ROOT; cd /tmp && export TMPDIR=/tmp
ARGS="--batch 64 --level 3 --launches 4 --sigma $1"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS --kernel-include-regex k_track_gn --output-format csv -d /tmp/pmc1 -- python3 $R/tools/probe_gn.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --kernel-include-regex k_track_gn --output-format csv -d /tmp/pmc2 -- python3 $R/tools/probe_gn.py $ARGS > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_WAIT_INST_LDS SQ_IFETCH SQ_INST_LEVEL_VMEM --kernel-include-regex k_track_gn --output-format csv -d /tmp/pmc3 -- python3 $R/tools/probe_gn.py $ARGS > /dev/null 2>&1
PMC_LAST=4 python3 $R/tools/pmc_summary.py "/tmp/pmc1/*/*counter_collection.csv" "/tmp/pmc2/*/*counter_collection.csv" "/tmp/pmc3/*/*counter_collection.csv" | grep -A26 "grid=622592"
