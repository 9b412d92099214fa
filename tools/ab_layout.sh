#!/bin/bash
# Library variants side by side, fixed work: value and the per-kind kernel milliseconds.  Usage: tools/ab_layout.sh <variant>...  -> gpurun_out/ab_layout.txt
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/ab_layout.txt; : > "$out"
for rep in 1 2 3; do
    for v in default "$@"; do
        lib=$root/lut_ldpc_amd/lib/liblut_ldpc_amd.so
        [ "$v" != default ] && lib=$root/lut_ldpc_amd/lib_variants/$v/liblut_ldpc_amd.so
        LUTLDPC_LIB=$lib python "$root/bench.py" --no-cpu-baseline --frame-loop-steps 0 --as-shipped-steps 0 --reps 0 --steps 5 --workload ${WL:-dvbs2} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('${WL:-dvbs2} $v: %.1f k cw/s' % (d['value']/1e3), 'ms/step %.2f' % d['ms_per_step'], 'kernels', {k: round(x, 3) for k, x in d['kernel_ms_per_step'].items()})" >> "$out"
        tail -1 "$out"
    done
done
