#!/bin/bash
# runs the CLI many times on a few random scenes; stops at the first crash
cd tests && PYTHONPATH=.. python3 -c "
import mesh_scene, pathlib
for s in (0,1,2,5,9): print(mesh_scene.write_random_scene(pathlib.Path('/tmp'),s))" && cd ..
fails=0
for i in $(seq 1 40); do
  for s in 0 1 2 5 9; do
    RT_AMD_BACKTRACE=1 timeout -k 5 60 ./raytracer_amd/rt_amd /tmp/random_$s.xml /tmp/o_$s.png > /tmp/cli.out 2> /tmp/cli.err
    rc=$?
    if [ $rc -ne 0 ]; then echo "run $i seed $s rc=$rc"; cat /tmp/cli.err; fails=$((fails+1)); [ $rc -ge 124 ] && exit 1; fi
  done
done
echo "fails=$fails"
