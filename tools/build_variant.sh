#!/bin/bash
# developer A/B: tools/build_variant.sh <name> [-DRT_...=...]...  ->  ab/<name>.so (librt_hip.so built with extra defines)
# run the variants on the GPU box with tools/ab.sh
set -e
NAME=$1; shift
mkdir -p ab build/ab_$NAME
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall"
for f in rt_api rt_kernels; do
  /opt/rocm/bin/hipcc $FLAGS "$@" -c -o build/ab_$NAME/$f.o raytracer_amd/csrc/$f.hip &
done
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o ab/$NAME.so build/ab_$NAME/rt_api.o build/ab_$NAME/rt_kernels.o build/rt_bvh.o build/rt_photon_order.o
echo built ab/$NAME.so
