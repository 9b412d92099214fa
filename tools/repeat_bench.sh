#!/bin/bash
# tools/repeat_bench.sh <out-file> <repetitions> <bench args...> : value / fused ms of repeated identical runs
OUT=$1; N=$2; shift 2
for i in $(seq 1 $N); do
    python3 bench.py --no-cpu-baseline --frame-loop-steps 0 --as-shipped-steps 0 "$@" 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*', round(r['value']), 'cw/s', round(r['kernel_ms_per_step'].get('fused_pass',0),2), 'ms fused', round(r['roofline_whole_decode']['device_copy_GBps_measured']), 'GB/s copy')" >> "$OUT"
done
