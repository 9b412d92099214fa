#!/bin/bash
# tools/sweep_args.sh <out-file> "<bench args>" ...   : one bench.py run per argument string
OUT=$1; shift
: > "$OUT"
for cfg in "$@"; do
    echo "== $cfg" >> "$OUT"
    python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline $cfg 2>>"$OUT.err" | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(round(r['value']), 'cw/s', round(r['ms_per_step'],2), 'ms/step', 'roof', r['roofline']['achieved'] and round(r['roofline']['achieved'] if r['roofline']['achieved']==r['roofline']['achieved'] else 0), r['roofline']['kernel'][:20], {k:round(v,2) for k,v in r['kernel_ms_per_step'].items()}, 'iters', round(r['config']['mean_iterations_executed'],1), 'frame_loop', round(r.get('frame_loop',{}).get('codewords_per_s_per_gpu',0)), 'Gbit/s', round(r['coded_bits_per_s']/1e9,2))" >> "$OUT" || echo FAILED >> "$OUT"
done
cat "$OUT"
