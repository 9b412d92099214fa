#!/bin/bash
# As-shipped mode at several SNRs with compaction off / automatic / stricter cost margins (run on the GPU box):
#   tools/shipped_policy_probe.sh [snr ...]  ->  gpurun_out/shipped_policy.txt
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/shipped_policy.txt; : > "$out"
for snr in ${@:-1.51 3.0 4.0}; do
  for env in "LUTLDPC_COMPACT=0" "X=0" "LUTLDPC_COMPACT_MARGIN=2" "LUTLDPC_COMPACT_MARGIN=4"; do
    env $env python "$root/bench.py" --no-cpu-baseline --no-configs --frame-loop-steps 0 --as-shipped-steps 0 --reps 0 --steps 6 --mode shipped --snr $snr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('snr $snr $env: %.1f k cw/s' % (d['value']/1e3), 'iters %.2f' % d['config']['mean_iterations_executed'], 'ms/step %.2f' % d['ms_per_step'], 'kernels', {k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()})" >> "$out"
    tail -1 "$out"
  done
done
