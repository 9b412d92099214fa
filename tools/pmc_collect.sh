#!/bin/bash
# Collects SQ / LDS / L2 counters of the decode kernels with rocprofv3 (one --pmc pass per counter group,
# --kernel-trace only, as the pool requires).  Usage: tools/pmc_collect.sh <out-dir> [bench args...]
set -e
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # name, counters...
    local name=$1; shift
    rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$OUT/$name" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events $BENCH_ARGS > "$R/$OUT/$name.log" 2>&1
}
run sq_a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq_b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
run sq_c SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE
find "$R/$OUT" -name "*counter_collection.csv" | head
