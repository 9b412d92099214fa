#!/bin/bash
# every bench workload in both modes, with the oracle check on a sample (cpu_baseline leg): one line each
OUT=$1; : > "$OUT"
for w in dvbs2 twin c2 c1 c5 c5chk; do
  for m in fixed shipped; do
    python3 bench.py --workload $w --mode $m --steps 4 --warmup 2 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=r['cpu_baseline']
print('$w $m', round(r['value']), 'cw/s', round(r['coded_bits_per_s']/1e9,2), 'Gbit/s', 'iters', round(r['config']['mean_iterations_executed'],1), 'frame_loop', round(r['frame_loop']['codewords_per_s_per_gpu']), 'cpu', round(c['value'],2), 'match', c['gpu_matches_oracle_on_sample'], 'roof', round(r['roofline']['achieved'] or 0))" >> "$OUT" || echo "$w $m FAILED" >> "$OUT"
  done
done
cat "$OUT"
