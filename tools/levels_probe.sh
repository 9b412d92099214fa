#!/bin/bash
# Repeats a counter-collection run: GRBM_GUI_ACTIVE (cycles) and durations of the fused kernel per process
OUT=$1; N=$2
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d "$R/$OUT/run$i" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --frame-loop-steps 0 --as-shipped-steps 0 > "$R/$OUT/run$i.json" 2>/dev/null
done
