#!/bin/bash
# PMC passes for one bench workload (run on the GPU box): tools/pmc.sh <workload> <outdir>
# Each pass is its own run (rocprofv3 does not split counters over passes).
W=${1:-c2}
OUT=${2:-$GRAFT_REPO_ROOT/gpurun_out/pmc_$W}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  timeout -s KILL 120 rocprofv3 --pmc "$@" -d $OUT/$name -o $name --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extra > $OUT/$name.log 2>&1 || return 1
}
run sq1 SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_INSTS_LDS && \
run sq2 SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_IFETCH && \
run tcc1 TCC_HIT_sum TCC_MISS_sum && \
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum && \
run fetch FETCH_SIZE && \
run write WRITE_SIZE && \
run grbm GRBM_GUI_ACTIVE GRBM_COUNT
