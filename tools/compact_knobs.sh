#!/bin/bash
# As-shipped mode: sweep of the compaction cost-model knobs on one box.  Usage: tools/compact_knobs.sh [workload]  -> gpurun_out/compact_knobs.txt
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/compact_knobs.txt; : > "$out"
wl=${1:-dvbs2}
run() {
    env "$@" python "$root/bench.py" --no-cpu-baseline --frame-loop-steps 0 --reps 0 --steps 5 --workload $wl --mode shipped 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $*: %.1f k cw/s' % (d['value']/1e3), 'ms/step %.2f' % d['ms_per_step'], 'layout %.2f fused %.2f' % (d['kernel_ms_per_step']['layout'], d['kernel_ms_per_step']['fused_pass']))" >> "$out"
    tail -1 "$out"
}
run X=1
for m in 0.5 0.7; do for s in 0.15 0.25 0.35; do run LUTLDPC_COMPACT_MARGIN=$m LUTLDPC_COMPACT_MIN_SHARE=$s; done; done
run LUTLDPC_COMPACT_MARGIN=1.0 LUTLDPC_COMPACT_MIN_SHARE=0.25
run LUTLDPC_COMPACT_MARGIN=1.0 LUTLDPC_COMPACT_MIN_SHARE=0.15
run LUTLDPC_COMPACT_MARGIN=0.7 LUTLDPC_COMPACT_MIN_SHARE=0.25 LUTLDPC_COMPACT_EVERY=1
run LUTLDPC_COMPACT_MARGIN=0.7 LUTLDPC_COMPACT_MIN_SHARE=0.25 LUTLDPC_COMPACT_EVERY=3
run X=1
