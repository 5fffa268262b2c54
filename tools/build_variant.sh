#!/bin/bash
# Build a variant of libdvo.so from an alternative dvo_kernels.hip (or extra -D flags) for A/B timing on one GPU box:
#   bash tools/build_variant.sh ab/name.so [path/to/dvo_kernels.hip] [-DFLAG ...]
# The other translation units come from direct-visual-odometry_amd/build (run make first).
set -e
cd "$(dirname "$0")/.."
OUT=$1; shift
SRC=direct-visual-odometry_amd/csrc/dvo_kernels.hip
if [ -n "$1" ] && [ "${1:0:1}" != "-" ]; then SRC=$1; shift; fi
B=direct-visual-odometry_amd/build
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt \
  -w -Idirect-visual-odometry_amd/csrc "$@" -x hip -c $SRC -o /tmp/variant_$$.o
hipcc -shared -fPIC --offload-arch=gfx950 -o $OUT /tmp/variant_$$.o $B/dvo_map_kernels.o $B/dvo_mono.o $B/dvo_engine.o $B/dvo_capi.o $B/dvo_io.o $B/dvo_eval.o $B/dvo_store.o -lz -ldl
rm -f /tmp/variant_$$.o
