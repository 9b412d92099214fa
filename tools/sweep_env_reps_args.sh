#!/bin/bash
# tools/sweep_env_reps_args.sh <out-file> <reps> "<bench args>" "<ENV=..>" ... : like sweep_env_reps.sh with extra bench arguments
OUT=$1; N=$2; ARGS=$3; shift 3
for cfg in "$@"; do
    vals=""
    for i in $(seq 1 $N); do
        v=$(env $cfg python3 bench.py --no-cpu-baseline --frame-loop-steps 0 --as-shipped-steps 0 --no-kernel-events --steps 5 --warmup 2 $ARGS 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(r['value']/1000,1))")
        vals="$vals $v"
    done
    echo "$ARGS | $cfg :$vals" >> "$OUT"
done
