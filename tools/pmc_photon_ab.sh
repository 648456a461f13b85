#!/bin/bash
# Developer A/B (GPU box): SQ and TCP counters of photon_kernel on the C5 bench workload, RT_PHOTON_FORK=1 against 0
# (the switch of tools/experiments/r03_photon_fork.patch, measured and not kept: apply the patch first).
cd /tmp && export TMPDIR=/tmp
for f in 1 0; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_photon_fork$f
  mkdir -p $OUT
  run() {
    name=$1; shift
    RT_PHOTON_FORK=$f timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/$name.log 2>&1 || return 1
  }
  run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS && \
  run sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_IFETCH && \
  run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum && \
  run grbm GRBM_GUI_ACTIVE GRBM_COUNT || exit 1
  echo "== RT_PHOTON_FORK=$f"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT | awk '/== void rtd::photon_kernel<false, false>/{p=1;next} /^== /{p=0} p'
done
