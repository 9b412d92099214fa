#!/bin/bash
# Collects the artifacts behind the roofline numbers of one round (run on the GPU box through gpurun):
#   <out>/bench.json                 the default bench line
#   <out>/kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   <out>/pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv   HBM traffic, separate --pmc passes (kernel trace only)
#   <out>/pmc_sq_a.csv, pmc_sq_b.csv what limits the fused kernel on the chip (VALU issue, LDS, waits)
# Usage: [BENCH_ARGS="--workload c2 --batch 4096"] tools/profile_round.sh <out-dir> [the same extra bench args]
# (BENCH_ARGS reaches the counter passes, "$@" the bench line and the kernel trace: give both for another workload)
set -e
OUT=$1; shift
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
python3 "$R/bench.py" "$@" > "$R/$OUT/bench.json" 2> "$R/$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -- python3 "$R/bench.py" --no-cpu-baseline --no-configs --frame-loop-steps 0 --as-shipped-steps 0 --reps 0 "$@" > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/trace.err"
find "$R/$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$R/$OUT/kernel_stats.csv" \;
pmc() {   # name, counters...
    local name=$1; shift
    # (LUTLDPC_PLACE=0: the counter means are per launch of the decode proper -- the placement search of a large batch would add its
    # probe launches, zeroed rows and three iterations each, to the same kernel name; bytes and instructions per launch do not depend
    # on where the rows landed)
    LUTLDPC_PLACE=0 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$R/$OUT/pmc_$name" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --no-configs --frame-loop-steps 0 --as-shipped-steps 0 --no-kernel-events --reps 0 $BENCH_ARGS > /dev/null 2> "$R/$OUT/pmc_$name.err"
    find "$R/$OUT/pmc_$name" -name "*counter_collection.csv" -exec cp {} "$R/$OUT/pmc_$name.csv" \;
    rm -rf "$R/$OUT/pmc_$name"
}
pmc FETCH_SIZE FETCH_SIZE
pmc WRITE_SIZE WRITE_SIZE
pmc sq_a GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY
pmc sq_b GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM
rm -rf "$R/$OUT/trace"
ls -la "$R/$OUT"
