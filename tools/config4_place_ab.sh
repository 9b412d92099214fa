#!/bin/bash
# BASELINE config 4 on one GPU (C++ ber_sim, one lane) with and without the placement search of the row buffers: what the search
# costs a run of this length against what it gains.  Usage (GPU box): tools/config4_place_ab.sh <out-dir under gpurun_out>
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/$1
mkdir -p "$OUT/base/codes"
cp "$R/data/codes/rate0.50_irreg_dvbs2_N64800.alist" "$OUT/base/codes/"
export LUTLDPC_DESIGN_CACHE=$R/data/design_cache
EXE=$R/lut_ldpc_amd/lib/ber_sim
"$EXE" -p "$R/data/params/ber.ini.dvbs2_sweep" -b "$OUT/base" -s 1 -c warm --lanes 1 > "$OUT/warm.log" 2>&1
for round in 1 2 3; do
  for v in default "LUTLDPC_PLACE=0"; do
    if [ "$v" = default ]; then e=LUTLDPC_DEBUG_ADDR=1; else e="$v LUTLDPC_DEBUG_ADDR=1"; fi
    t0=$(date +%s.%N)
    env $e "$EXE" -p "$R/data/params/ber.ini.dvbs2_sweep" -b "$OUT/base" -s 7 -c ab --lanes 1 > "$OUT/run.log" 2>&1
    t1=$(date +%s.%N)
    echo "$v round $round: wall $(python3 -c "print('%.2f' % ($t1 - $t0))") s; $(tail -n 1 "$OUT/run.log"); $(grep -c 'lutldpc placement' "$OUT/run.log") placement lines"
    grep 'lutldpc placement' "$OUT/run.log" | cut -c1-220 | sed 's/^/      /'
  done
done
rm -rf "$OUT/base"
