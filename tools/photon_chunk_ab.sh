#!/bin/bash
# Developer A/B (GPU box): photon_kernel's time against the records a wave reserves per atomic (ab/dc*.so, tools/build_variant.sh dcN -DRT_DEPOSIT_CHUNK=N)
cd /tmp && export TMPDIR=/tmp
for lib in raytracer_amd/librt_hip.so ab/dc64.so ab/dc128.so ab/dc256.so; do
  [ -f $GRAFT_REPO_ROOT/$lib ] || continue
  n=$(basename $lib .so)
  PHOTON_RECORDS=1048576 RT_HIP_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pc_$n -o ph --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/photon_profile.py > $GRAFT_REPO_ROOT/gpurun_out/pc_$n.log 2>&1 || exit 1
  echo "== $lib: $(grep ordered $GRAFT_REPO_ROOT/gpurun_out/pc_$n.log | cut -c1-60)"
  python3 -c "
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/pc_$n/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'photon_kernel' in r['Name'] or 'sum_runs' in r['Name'] or 'scatter' in r['Name'] or 'place' in r['Name']: print('  ', r['Name'][:50].ljust(52), r['Calls'].rjust(4), '%9.1f us' % (float(r['AverageNs'])/1e3))
"
done
