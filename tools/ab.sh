# A/B of library builds on the same box: bash tools/ab.sh ab/a.so ab/b.so ...   ("-" = the in-tree build)
for rep in 1 2; do
  for lib in "$@"; do
    if [ "$lib" = "-" ]; then p=""; else p="$PWD/$lib"; fi
    DVO_LIB_PATH=$p python bench.py --no-cpu-baseline --pcie-steps 0 --no-roofline | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']), round(d['ms_per_step'],3))"
  done
done
