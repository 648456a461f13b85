#!/bin/bash
# developer tool (GPU box): tools/hip_init_race.sh [runs]  — builds tools/hip_init_race and runs it `runs` times in each mode
RUNS=${1:-150}
OUT=gpurun_out/hip_init_race; mkdir -p $OUT
/opt/rocm/bin/hipcc -O2 -std=c++17 -pthread -o $OUT/hip_init_race tools/hip_init_race.cpp || exit 1
for mode in ${MODES:-overlap sequential steady create}; do
  fails=0
  for i in $(seq 1 $RUNS); do
    timeout -k 5 60 $OUT/hip_init_race $mode ${ROUNDS:-30} > $OUT/out.txt 2> $OUT/err.txt
    rc=$?
    if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "== $mode run $i rc=$rc" >> $OUT/failures.txt; grep -v amdgpu.ids $OUT/err.txt >> $OUT/failures.txt; [ $rc -ge 124 ] && [ $rc -ne 139 ] && exit 1; fi
  done
  echo "$mode: $fails failures of $RUNS runs"
done
[ -f $OUT/failures.txt ] && head -c 6000 $OUT/failures.txt
exit 0
