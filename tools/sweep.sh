#!/bin/bash
# tools/sweep.sh <out-file> "<ENV1=a ENV2=b>" ...   : one bench.py run per environment string
OUT=$1; shift
: > "$OUT"
for cfg in "$@"; do
    echo "== $cfg" >> "$OUT"
    env $cfg python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(r['value']), 'cw/s', round(r['ms_per_step'],2), 'ms/step', 'roof', round(r['roofline']['achieved']), {k:round(v,2) for k,v in r['kernel_ms_per_step'].items()})" >> "$OUT" || echo FAILED >> "$OUT"
done
cat "$OUT"
