#!/bin/bash
# As-shipped mode (parity_check_iter = true) with compaction of the surviving frames forced off / automatic / forced on, per
# workload, beside the fixed-work number of the same box.  Usage: tools/shipped_ab.sh [workload...]   -> gpurun_out/shipped_ab.txt
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/shipped_ab.txt; : > "$out"
run() {   # label, env..., -- bench args
    local label=$1; shift
    local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
    env "${envs[@]}" python "$root/bench.py" --no-cpu-baseline --frame-loop-steps 0 --as-shipped-steps 0 --reps 0 --steps 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$label', '%.1f k cw/s' % (d['value']/1e3), 'iters %.1f' % d['config']['mean_iterations_executed'], 'ms/step %.2f' % d['ms_per_step'])" >> "$out"
    tail -1 "$out"
}
for wl in ${@:-dvbs2 twin c2 c1}; do
    for rep in 1 2; do
        run "$wl fixed" X=1 -- --workload $wl --mode fixed
        run "$wl shipped,compaction-off" LUTLDPC_COMPACT=0 -- --workload $wl --mode shipped
        run "$wl shipped,default" X=1 -- --workload $wl --mode shipped
        run "$wl shipped,compaction-on" LUTLDPC_COMPACT=1 -- --workload $wl --mode shipped
    done
done
