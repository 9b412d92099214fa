#!/bin/bash
# Collects the artifacts behind the roofline numbers of one round (run on the GPU box through gpurun):
#   <out>/bench.json                 the default bench line
#   <out>/kernel_stats.csv           rocprofv3 --kernel-trace --stats of the same command
#   <out>/pmc_fetch.csv, pmc_write.csv   FETCH_SIZE / WRITE_SIZE, separate --pmc passes (kernel trace only)
# Usage: tools/profile_round.sh <out-dir>
set -e
OUT=$1
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
python3 "$R/bench.py" > "$R/$OUT/bench.json" 2> "$R/$OUT/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -- python3 "$R/bench.py" --no-cpu-baseline --frame-loop-steps 0 > "$R/$OUT/bench_under_rocprof.json" 2> "$R/$OUT/trace.err"
find "$R/$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$R/$OUT/kernel_stats.csv" \;
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$R/$OUT/pmc_$c" -- python3 "$R/bench.py" --steps 1 --warmup 1 --no-cpu-baseline --frame-loop-steps 0 --no-kernel-events > /dev/null 2> "$R/$OUT/pmc_$c.err"
    find "$R/$OUT/pmc_$c" -name "*counter_collection.csv" -exec cp {} "$R/$OUT/pmc_$c.csv" \;
done
rm -rf "$R/$OUT/trace" "$R/$OUT/pmc_FETCH_SIZE" "$R/$OUT/pmc_WRITE_SIZE"
ls -la "$R/$OUT"
