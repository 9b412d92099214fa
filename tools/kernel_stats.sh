#!/bin/bash
# rocprofv3 kernel trace + stats of one bench.py run.  Usage: tools/kernel_stats.sh <out-dir> [bench args]
set -e
OUT=$1; shift
mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/trace" -- python3 "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline "$@" > "$R/$OUT/bench.json" 2> "$R/$OUT/bench.err"
find "$R/$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$R/$OUT/kernel_stats.csv" \;
head -12 "$R/$OUT/kernel_stats.csv"
