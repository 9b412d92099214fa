#!/bin/bash
# As-shipped mode: finished frames keep their rows through a permutation (default) against the earlier flow (LUTLDPC_COMPACT_KEEP=0),
# and the cost-model knobs around the default.  Usage: tools/compact_keep.sh [workload]  -> gpurun_out/compact_keep.txt
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/compact_keep.txt; : > "$out"
wl=${1:-dvbs2}
run() {
    env "$@" python "$root/bench.py" --no-cpu-baseline --frame-loop-steps 0 --as-shipped-steps 0 --reps 0 --steps 5 --workload $wl --mode shipped 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $*: %.1f k cw/s' % (d['value']/1e3), 'ms/step %.2f' % d['ms_per_step'], 'layout %.2f fused %.2f' % (d['kernel_ms_per_step']['layout'], d['kernel_ms_per_step']['fused_pass']))" >> "$out"
    tail -1 "$out"
}
for rep in 1 2; do
run X=1
run LUTLDPC_COMPACT_KEEP=0
done
for m in 0.5 0.7 1.5; do run LUTLDPC_COMPACT_MARGIN=$m; done
for s in 0.15 0.25; do run LUTLDPC_COMPACT_MIN_SHARE=$s; run LUTLDPC_COMPACT_MIN_SHARE=$s LUTLDPC_COMPACT_MARGIN=0.7; done
run X=1
