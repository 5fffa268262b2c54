# A/B of library builds on the mono batch workload (mapping kernels): bash tools/ab_mono.sh ab/a.so - ...   ("-" = the in-tree build)
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then p=""; else p="$PWD/$lib"; fi
    DVO_LIB_PATH=$p python bench.py --workload syn640-mono --no-cpu-baseline | python -c "
import json,sys
d=json.loads(sys.stdin.read()); mk=d.get('mapping_kernels',{})
print('$lib', round(d['value']), round(d['ms_per_step'],3), 'update_us', round(mk.get('k_depth_update',{}).get('avg_launch_us',0),1), 'regdec_us', round(mk.get('k_regularize_redecimate',{}).get('avg_launch_us',0),1))"
  done
done
