#!/bin/bash
# Small batches whose rows fit the 256 MB Infinity Cache, default cache policy against non-temporal rows (run on the GPU box):
#   tools/build_variant.sh cache_default "-DLUTLDPC_LD_AUX=0 -DLUTLDPC_ST_AUX=0"; gpurun -- tools/mall_probe.sh gpurun_out/mall
OUT=${1:-gpurun_out/mall}; mkdir -p "$OUT"
R=$GRAFT_REPO_ROOT
for B in 512 1024 1536 2048 4096 32768; do
  for V in lib lib_variants/cache_default; do
    [ -f "$R/lut_ldpc_amd/$V/liblut_ldpc_amd.so" ] || continue
    steps=$(( 32768 / B * 3 )); [ $steps -lt 6 ] && steps=6
    LUTLDPC_LIB=$R/lut_ldpc_amd/$V/liblut_ldpc_amd.so python3 "$R/bench.py" --batch $B --steps $steps --warmup 3 --no-cpu-baseline --no-configs --frame-loop-steps 0 --as-shipped-steps 0 --reps 0 \
      > "$OUT/b${B}_$(basename $V).json" 2> "$OUT/b${B}_$(basename $V).err" || { echo "failed B=$B $V"; tail -3 "$OUT/b${B}_$(basename $V).err"; }
    python3 - "$OUT/b${B}_$(basename $V).json" $B $V <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().split('\n')[-1])
print(f"B={sys.argv[2]:>6} {sys.argv[3]:<28} {d['value']/1e3:8.1f} k cw/s  fused launch {d['roofline']['avg_launch_ms']*1e3:8.1f} us  frac {d['roofline']['frac']:.3f}", flush=True)
PY
  done
done | tee "$OUT/summary.txt"
