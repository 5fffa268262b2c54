for rep in 1 2; do for lib in "$@"; do echo -n "$lib  "; DVO_LIB_PATH=$PWD/$lib python tools/probe_gn.py --batch 256 --level 3 --launches 20 --sigma 0.5 | tail -1; done; done
