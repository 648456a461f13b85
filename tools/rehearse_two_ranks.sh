#!/bin/bash
# developer tool (GPU box with ONE GPU): the torch driver with two ranks sharing the GPU, collectives through gloo
# (RT_DIST_BACKEND=gloo), on small variants of the caustic animation (plain and motion blur) and a Julia still: the files must
# be the single-process ones, byte for byte.  Two processes on one GPU take turns at it, so this is slow for what it renders.
# (Round 2 noted that bench.py --gpus 2 over gloo "took 1 s per 1.8 ms step when it ran at all" and blamed the two grids
# sharing the GPU.  It was torch's gloo backend: handed a device tensor while kernels are in flight it hangs in Work.wait(),
# with one rank and torch kernels alone too — tools/gloo_cuda_gather_probe.py, DESIGN.md 5.  driver.settle_for_gloo waits for
# the device before such a call; since then `RT_DIST_BACKEND=gloo torchrun --nproc-per-node 2 bench.py --gpus 2` runs.)
set -e
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/rehearse; mkdir -p $OUT
S=scenes/examples3
sed -e 's/horizontal="300" vertical="300"/horizontal="48" vertical="40"/' -e 's/steps="1000"/steps="120"/' -e 's/length="2" fps="10"/length="1" fps="5"/' $S/103_caustic_animation.xml > $S/_rehearse_anim.xml
sed -e 's/horizontal="300" vertical="300"/horizontal="48" vertical="40"/' -e 's/steps="1000"/steps="120"/' -e 's/length="2" fps="10"/length="1" fps="3"/' $S/103_caustic_animation_mb8.xml > $S/_rehearse_mb.xml
sed -e 's/horizontal="960" vertical="600"/horizontal="96" vertical="60"/' $S/101_julia_shiny.xml > $S/_rehearse_julia.xml
trap 'rm -f $S/_rehearse_*.xml' EXIT
PORT=29540
for n in anim mb julia; do
  PORT=$((PORT + 1))
  echo "== $n: one process"
  timeout -k 10 100 python -m raytracer_amd.driver $S/_rehearse_$n.xml $OUT/one_$n.png
  echo "== $n: two ranks"
  RT_DIST_BACKEND=gloo timeout -k 10 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT \
      -m raytracer_amd.driver $S/_rehearse_$n.xml $OUT/two_$n.png 2>&1 | grep -v "amdgpu.ids\|socket.cpp\|Gloo" || true
  cmp $OUT/one_$n.png $OUT/two_$n.png && echo "$n: same file"
done
