#!/bin/bash
# Developer A/B (GPU box): per-kernel times of twenty C5 photon passes with and without RT_PHOTON_FORK
# (the switch of tools/experiments/r03_photon_fork.patch, measured and not kept: apply the patch first).
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  RT_PHOTON_FORK=$f timeout -k 10 200 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ph_prof$f -o ph --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/photon_profile.py > $GRAFT_REPO_ROOT/gpurun_out/ph_prof$f.log 2>&1 || exit 1
  echo "== RT_PHOTON_FORK=$f"
  python3 -c "
import csv,glob,os
f=glob.glob(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ph_prof$f/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)): print(r['Name'][:60].ljust(62), r['Calls'].rjust(4), '%9.1f us' % (float(r['AverageNs'])/1e3))
"
done
