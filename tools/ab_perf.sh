#!/bin/bash
# developer A/B (GPU box): the four timing scenes and two parity scenes for the in-tree build and every ab/*.so
# usage: tools/ab_perf.sh [reps]
R=${1:-5}
bash tools/ab.sh "scenes/examples/example9_4k.xml --no-oracle --reps $R" "scenes/examples2/8_fresnel_ss4.xml --no-oracle --reps $R" \
  "scenes/examples3/101_julia_shiny.xml --no-oracle --reps $R" "scenes/examples3/103_caustic_animation.xml --t 0.5 --no-oracle --reps $R" \
  "scenes/examples2/8_fresnel.xml --reps 1" "scenes/examples/example9_lowres.xml --reps 1" 2>&1 | grep -v amdgpu.ids
