cd $GRAFT_REPO_ROOT; export DVO_LIB_PATH=$PWD/ab/libdvo_clean.so
for rep in 1 2; do
  echo -n "probe clean ON : "; DVO_EXP_CLEAN_ON=1 python3 tools/probe_gn.py --raw --batch 256 --level 3 --launches 20 --sigma 0.1 --ppt 4 --group 2 2>/dev/null | tail -1
  echo -n "probe clean OFF: "; python3 tools/probe_gn.py --raw --batch 256 --level 3 --launches 20 --sigma 0.1 --ppt 4 --group 2 2>/dev/null | tail -1
done
for v in ON OFF; do
  if [ $v = ON ]; then export DVO_EXP_CLEAN_ON=1; else unset DVO_EXP_CLEAN_ON; fi
  python3 bench.py --no-cpu-baseline --pcie-steps 0 --no-secondary --batch 4096 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench clean $v:', round(d['value']), 'frames/s', round(d['ms_per_step'],3), 'ms/step, gn avg', round(d['roofline']['avg_launch_us'],1), 'us', d['accuracy']['per_iteration_parity']['ok'] if 'accuracy' in d and 'per_iteration_parity' in d['accuracy'] else '')"
done
