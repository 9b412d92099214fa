#!/bin/bash
# A/B of library variants built by tools/build_variant.sh, interleaved (default, variants..., default, variants...) so that
# the run-to-run levels of a box show up as spread within each variant.  Usage: tools/ab_variants.sh <rounds> <variant>...
# Output: gpurun_out/ab_variants.txt (value k cw/s, fused launch ms per run), then medians per variant
rounds=${1:-2}; shift
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out; mkdir -p "$out"
: > "$out/ab_variants.txt"
for r in $(seq 1 "$rounds"); do
  for v in default "$@"; do
    # a variant is  <build>[:ENV=VAL[,ENV=VAL...]]  -- <build> = default or a directory of lut_ldpc_amd/lib_variants/
    b=${v%%:*}; envs=""; if [ "$b" != "$v" ]; then envs=$(echo "${v#*:}" | tr ',' ' '); fi
    if [ "$b" = default ]; then lib=$root/lut_ldpc_amd/lib/liblut_ldpc_amd.so; else lib=$root/lut_ldpc_amd/lib_variants/$b/liblut_ldpc_amd.so; fi
    env $envs LUTLDPC_LIB=$lib python "$root/bench.py" --no-cpu-baseline --frame-loop-steps 0 --as-shipped-steps 0 --steps 5 ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
ss=d.get('single_steps') or {}
pc=(ss.get('power_clock') or {})
print('$v', 'round $r', 'k_cw_s %.1f' % (d['value']/1e3), 'fused_ms %.4f' % d['roofline']['avg_launch_ms'], 'frac %.3f' % d['roofline']['frac'], 'step_ms med/min/max %.2f %.2f %.2f' % (ss.get('median_ms',0), ss.get('min_ms',0), ss.get('max_ms',0)), 'power', pc.get('socket_power_W'), 'sclk', pc.get('sclk_MHz'))
" >> "$out/ab_variants.txt"
    tail -1 "$out/ab_variants.txt"
  done
done
python3 - "$out/ab_variants.txt" <<'PY'
import sys, collections, statistics
v = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    p = ln.split()
    if "k_cw_s" in p:
        v[p[0]].append(float(p[p.index("k_cw_s") + 1]))
print("---- medians (k cw/s): " + "  ".join(f"{k}: {statistics.median(x):.1f} [{min(x):.1f}..{max(x):.1f}] n={len(x)}" for k, x in v.items()))
PY
