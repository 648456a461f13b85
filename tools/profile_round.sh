#!/bin/bash
# Collects the round's profiles on the GPU box: tools/profile_round.sh <tag> [workloads...]
# For each workload: rocprofv3 --kernel-trace --stats of the bench command, then the PMC passes
# (tools/pmc.sh, one run per counter group).  Summaries land in gpurun_out/profiles_<tag>/ ; copy what
# is to be kept into profiles/.
TAG=${1:-r01}; shift
WL=${@:-c2}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/profiles_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in $WL; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_$W -o $W --output-format csv -- python3 $ROOT/bench.py --workload $W --steps 10 --warmup 2 --no-cpu-baseline --no-extra > $OUT/bench_$W.json 2> $OUT/bench_$W.err || exit 1
  cp $(find $OUT/trace_$W -name "*kernel_stats.csv" | head -1) $OUT/${TAG}_${W}_kernel_stats.csv
  bash $ROOT/tools/pmc.sh $W $OUT/pmc_$W || exit 1
  python3 $ROOT/tools/pmc_summary.py $OUT/pmc_$W > $OUT/${TAG}_${W}_pmc_summary.txt
done
