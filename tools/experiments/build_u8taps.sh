#!/bin/bash
# EXPERIMENT build: libdvo with -DDVO_EXP_U8TAPS -> ab/libdvo_u8taps.so.  k_track_gn then takes the 12
# taps of the finest level from the reference's raw u8 frame (24 B per pixel through the L1 instead of 48, + 12 conversions and 12
# multiplications); DVO_EXP_U8TAPS_OFF=1 in the environment switches the same build back to float taps (the A side of the A/B).
# (apply tools/experiments/k_track_gn_u8taps.patch first: git apply tools/experiments/k_track_gn_u8taps.patch)
cd "$(dirname "$0")/../../direct-visual-odometry_amd" && mkdir -p ../ab/build_u8 &&
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-function -DDVO_EXP_U8TAPS"
for f in dvo_kernels.hip dvo_map_kernels.hip; do hipcc $FL -c csrc/$f -o ../ab/build_u8/${f%.*}.o & done
for f in dvo_mono dvo_engine dvo_capi dvo_io dvo_eval dvo_store; do hipcc $FL -x hip -c csrc/$f.cpp -o ../ab/build_u8/$f.o & done
wait
hipcc -shared -fPIC --offload-arch=gfx950 -o ../ab/libdvo_u8taps.so ../ab/build_u8/*.o -lz -ldl && ls -la ../ab/libdvo_u8taps.so
