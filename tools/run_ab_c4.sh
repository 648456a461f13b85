#!/bin/bash
# developer A/B (GPU box): the in-tree build and every ab/*.so on the Julia scene: a small frame against the oracle, then timing
# at 120x75 (the chain of the longest sample) and at full size, for each value of RT_JULIA_HOT_STEPS given as arguments
# (default: the library's own)
for lib in raytracer_amd/librt_hip.so ab/*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  RT_HIP_LIB=$PWD/$lib timeout -k 10 100 python tools/gpu_check.py scenes/examples3/101_julia_shiny.xml --width 240 --height 150 --reps 1 || exit 1
  for fs in ${@:-default}; do
    echo "-- RT_JULIA_HOT_STEPS=$fs"
    if [ "$fs" = default ]; then unset RT_JULIA_HOT_STEPS; else export RT_JULIA_HOT_STEPS=$fs; fi
    RT_HIP_LIB=$PWD/$lib timeout -k 10 100 python tools/gpu_check.py scenes/examples3/101_julia_shiny.xml --width 120 --height 75 --no-oracle --reps 3 || exit 1
    RT_HIP_LIB=$PWD/$lib timeout -k 10 100 python tools/gpu_check.py scenes/examples3/101_julia_shiny.xml --no-oracle --reps 5 || exit 1
  done
done
