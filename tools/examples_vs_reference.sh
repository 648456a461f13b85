#!/bin/bash
# Developer sweep (GPU box): every example scene the reference ships (scenes/examples*/ *.xml, 7_dof excepted: the
# reference's lens draws are not reproducible) through raytracer_amd/rt_amd and through the reference's own executable
# (oracle/_ref/raytracer_ostrict, built from the reference's sources in the build container): the files must be the same bytes.
# A scene whose reference run exceeds $1 seconds (default 170) is reported as skipped.
cd "$GRAFT_REPO_ROOT"
CAP=${1:-170}
OUT=gpurun_out/examples; mkdir -p $OUT
same=0; diff=0; skipped=0
for f in scenes/examples/*.xml scenes/examples2/*.xml scenes/examples3/*.xml; do
  b=$(basename $f .xml)
  case $b in 7_dof) echo "$b: skipped (depth of field)"; continue;; esac
  d=$(dirname $f)
  t0=$(date +%s%N)
  (cd $d && timeout -k 5 $CAP $GRAFT_REPO_ROOT/oracle/_ref/raytracer_ostrict $b.xml $GRAFT_REPO_ROOT/$OUT/ref_$b.png > /dev/null 2>&1)
  rc=$?
  t1=$(date +%s%N)
  if [ $rc -ne 0 ]; then echo "$b: reference run rc=$rc after $(( (t1 - t0) / 1000000 )) ms: skipped"; skipped=$((skipped+1)); continue; fi
  (cd $d && timeout -k 5 120 $GRAFT_REPO_ROOT/raytracer_amd/rt_amd $b.xml $GRAFT_REPO_ROOT/$OUT/hip_$b.png > /dev/null 2>&1) || { echo "$b: rt_amd failed"; diff=$((diff+1)); continue; }
  t2=$(date +%s%N)
  if cmp -s $OUT/ref_$b.png $OUT/hip_$b.png; then rm -f $OUT/ref_$b.png $OUT/hip_$b.png; same=$((same+1)); echo "$b: same file (reference $(( (t1 - t0) / 1000000 )) ms, rt_amd $(( (t2 - t1) / 1000000 )) ms, process start to exit)"; else diff=$((diff+1)); echo "$b: FILES DIFFER"; fi
done
echo "same $same, differing $diff, skipped $skipped"
