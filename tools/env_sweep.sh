#!/bin/bash
# Fixed-work bench under a list of environment settings on one box, alternating, two rounds.
# Usage: tools/env_sweep.sh <workload> "VAR=a" "VAR=b ..." ...   -> gpurun_out/env_sweep.txt   (use X=1 for the default)
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/env_sweep.txt; : > "$out"
wl=$1; shift
for rep in 1 2; do
    for e in "$@"; do
        env $e python "$root/bench.py" --no-cpu-baseline --frame-loop-steps 0 --as-shipped-steps 0 --reps 0 --no-configs --steps 5 --workload $wl 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl [$e]: %.1f k cw/s' % (d['value']/1e3), 'ms/step %.2f' % d['ms_per_step'], 'fused %.2f' % d['kernel_ms_per_step']['fused_pass'], 'chain', d['config']['kernels'].get('chain_nodes'), 'cn_epw', d['config']['kernels'].get('cn_edges_per_wave'))" >> "$out"
        tail -1 "$out"
    done
done
