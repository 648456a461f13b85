#!/bin/bash
# developer tool (no GPU needed): register, spill, scratch and LDS use of every kernel of rt_kernels.hip as the
# in-tree flags compile it (the metadata notes of the gfx950 code object).  usage: tools/kernel_resources.sh > profiles/rNN_kernel_resources.txt
set -e
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize "$@" -S --cuda-device-only \
    -o $TMP/k.s raytracer_amd/csrc/rt_kernels.hip 2>/dev/null
echo "kernel | VGPRs | spilled VGPRs | SGPRs | spilled SGPRs | scratch bytes per lane | static LDS bytes"
awk '/^    \.name:/ {name=$2} /\.group_segment_fixed_size:/ {lds=$2} /\.private_segment_fixed_size:/ {scr=$2} /\.sgpr_count:/ {sg=$2} /\.sgpr_spill_count:/ {sgs=$2}
     /\.vgpr_count:/ {vg=$2} /\.vgpr_spill_count:/ {vgs=$2} /\.wavefront_size:/ {print name" | "vg" | "vgs" | "sg" | "sgs" | "scr" | "lds}' $TMP/k.s | c++filt | sed 's/(rtd::DevScene, rtd::Dev[A-Za-z]*)//; s/^void //'
rm -rf $TMP
