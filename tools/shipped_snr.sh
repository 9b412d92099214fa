#!/bin/bash
# As-shipped mode over the SNR range of the config-4 sweep: decode-only and full frame-loop rates, mean iterations, kernel shares.
# Usage: tools/shipped_snr.sh [workload] [snr...]   -> gpurun_out/shipped_snr.txt
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/shipped_snr.txt; : > "$out"
wl=${1:-dvbs2}; shift
for snr in ${@:-1.11 1.5 2.0 3.0 4.0}; do
    python "$root/bench.py" --no-cpu-baseline --frame-loop-steps 2 --reps 0 --steps 5 --workload $wl --mode shipped --snr $snr 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl snr $snr: %.1f k cw/s' % (d['value']/1e3), 'iters %.2f' % d['config']['mean_iterations_executed'], 'ms/step %.2f' % d['ms_per_step'],
      'loop %.1f k cw/s' % (d['frame_loop']['codewords_per_s_per_gpu']/1e3), 'frontend %.2f ms' % d['frame_loop']['frontend_ms_per_step'],
      'kernels', {k: round(v, 2) for k, v in d['kernel_ms_per_step'].items()}, 'host-clock decode %.2f' % d['decode_ms_per_step_host_clock'])" >> "$out"
    tail -1 "$out"
done
