#!/bin/bash
# As-shipped mode, library variants side by side on the same box, alternating.  Usage: tools/ab_shipped.sh <workload> <variant|default>...
# -> gpurun_out/ab_shipped.txt
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/ab_shipped.txt; : > "$out"
wl=$1; shift
for rep in 1 2 3; do
    for v in "$@"; do
        lib=$root/lut_ldpc_amd/lib/liblut_ldpc_amd.so
        [ "$v" != default ] && lib=$root/lut_ldpc_amd/lib_variants/$v/liblut_ldpc_amd.so
        LUTLDPC_LIB=$lib python "$root/bench.py" --no-cpu-baseline --frame-loop-steps 0 --reps 0 --steps 5 --workload $wl --mode shipped 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$wl $v: %.1f k cw/s' % (d['value']/1e3), 'ms/step %.2f' % d['ms_per_step'], 'kernels', {k: round(x, 2) for k, x in d['kernel_ms_per_step'].items()})" >> "$out"
        tail -1 "$out"
    done
done
