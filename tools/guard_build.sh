#!/bin/bash
# Developer build with index guards in the photon pass (-DRT_GUARD: every index of photon_cull_kernel, photon_kernel and the
# bucket kernels of rt_photon_order.hip is checked, the first violation is reported by rt_photon_generate on stderr and
# the access skipped, so a wrong index shows without a memory fault on the GPU):  tools/guard_build.sh  ->  ab/guard.so
set -e
mkdir -p ab build/ab_guard
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wall -DRT_GUARD"
for f in rt_api rt_kernels rt_photon_order; do
  /opt/rocm/bin/hipcc $FLAGS -c -o build/ab_guard/$f.o raytracer_amd/csrc/$f.hip &
done
wait
/opt/rocm/bin/hipcc $FLAGS -shared -o ab/guard.so build/ab_guard/rt_api.o build/ab_guard/rt_kernels.o build/ab_guard/rt_photon_order.o build/rt_bvh.o
echo built ab/guard.so
