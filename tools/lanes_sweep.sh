#!/bin/bash
# Developer sweep (GPU box): units/s of the 160-unit C5 file against the driver's lanes and the HIP hardware queues.
# tools/lanes_sweep.sh "<lanes...>" "<queues...>"
cd $GRAFT_REPO_ROOT
for q in $2; do
  for n in $1; do
    GPU_MAX_HW_QUEUES=$q RT_DRIVER_LANES=$n timeout -k 10 150 python3 bench.py --workload c5mb --steps 5 --warmup 2 > gpurun_out/c5mb_sweep.json 2> gpurun_out/c5mb_sweep.err || exit 1
    python3 -c "
import json
d=json.load(open('gpurun_out/c5mb_sweep.json'))
print('queues $q lanes $n: %.1f ms per file, %.0f units/s, busy %s' % (d['ms_per_step'], d['config']['units_per_s'], d['config']['gpu_busy_frac_per_rank']))"
  done
done
