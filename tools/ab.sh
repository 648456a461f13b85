#!/bin/bash
# developer A/B: runs tools/gpu_check.py with the given argument sets for every ab/*.so and the in-tree build
# usage: tools/ab.sh "<gpu_check args>" ["<more args>" ...]   (on the GPU box)
for lib in raytracer_amd/librt_hip.so ab/*.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  for a in "$@"; do
    RT_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/gpu_check.py $a || exit 1
  done
done
