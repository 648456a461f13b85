#!/bin/bash
# developer A/B: runs tools/gpu_check.py on the given scenes for every ab/*.so and the in-tree build
# usage: tools/ab.sh "<gpu_check args>"   (on the GPU box)
for lib in raytracer_amd/librt_hip.so ab/*.so; do
  echo "== $lib"
  RT_HIP_LIB=$PWD/$lib timeout -k 10 200 python tools/gpu_check.py $1 || exit 1
done
