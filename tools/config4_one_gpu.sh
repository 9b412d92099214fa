#!/bin/bash
# BASELINE config 4 (data/params/ber.ini.dvbs2_sweep: DVB-S2 N=64800, Nframes = 1e6 per SNR point, SNRdB = 0:.5:4, parity_check_iter = true)
# end to end through the C++ `ber_sim` on the ONE GPU of a gpurun box: wall clock of the whole command, frames simulated per SNR point
# (the result file), and the same command under rocprofv3 --kernel-trace --stats.  The 8-GPU run of the same file is the driver's.
# Usage (on the GPU box): tools/config4_one_gpu.sh <out-dir under gpurun_out> [lanes ...]
set -e
R=$GRAFT_REPO_ROOT; OUT=$R/$1; shift; LANES=${@:-"1 2"}
mkdir -p "$OUT/base/codes"
cp "$R/data/codes/rate0.50_irreg_dvbs2_N64800.alist" "$OUT/base/codes/"
export LUTLDPC_DESIGN_CACHE=$R/data/design_cache
EXE=$R/lut_ldpc_amd/lib/ber_sim
"$EXE" -p "$R/data/params/ber.ini.dvbs2_sweep" -b "$OUT/base" -s 1 -c warm --lanes 1 > "$OUT/warm.log" 2>&1 || { tail -5 "$OUT/warm.log"; exit 1; }   # design cache, hiprtc, page-in
for l in $LANES; do
    t0=$(date +%s.%N)
    "$EXE" -p "$R/data/params/ber.ini.dvbs2_sweep" -b "$OUT/base" -s 7 -c lanes$l --lanes $l > "$OUT/run_lanes$l.log" 2>&1
    t1=$(date +%s.%N)
    python3 - "$OUT" $l $t0 $t1 "$R" <<'PY'
import sys, glob
sys.path.insert(0, sys.argv[5] + "/tests")
from itfile_reader import itload
out, lanes, t0, t1 = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
f = sorted(glob.glob(f"{out}/base/results/*lanes{lanes}/*_rseed0007.it"))[-1]
d = itload(f)
frames = [int(x) for x in d["sim_Nframes"]]
print(f"lanes {lanes}: wall {t1 - t0:.2f} s (process start, design from cache, hiprtc, sweep, result file)  frames per SNR point {frames}  total {sum(frames)}"
      f"  -> {sum(frames) / (t1 - t0) / 1e3:.1f} k codewords/s end to end")
for k in sorted(d):
    if k.startswith("sim_"):
        print("   ", k, [float(x) for x in d[k]])
PY
done
# the sharded Python driver (lut_ldpc_amd/ber_sim.py, the one torch.distributed launches on 8 GPUs) as ONE process on the same file and
# seed: its result file must hold the same counters as the C++ run
t0=$(date +%s.%N)
(cd "$R" && python3 -m lut_ldpc_amd.ber_sim -p "$R/data/params/ber.ini.dvbs2_sweep" -b "$OUT/base" -s 7 -c py > "$OUT/run_py.log" 2>&1) || { tail -5 "$OUT/run_py.log"; exit 1; }
t1=$(date +%s.%N)
python3 - "$OUT" $t0 $t1 "$R" <<'PY'
import sys, glob
import numpy as np
sys.path.insert(0, sys.argv[4] + "/tests")
from itfile_reader import itload
out, t0, t1 = sys.argv[1], float(sys.argv[2]), float(sys.argv[3])
py = itload(sorted(glob.glob(f"{out}/base/results/*py/*_rseed0007.it"))[-1])
cc = itload(sorted(glob.glob(f"{out}/base/results/*lanes1/*_rseed0007.it"))[-1])
keys = ("sim_SNRdB", "sim_Nframes", "sim_Ndatabits", "sim_frame_errors", "sim_data_bit_errors", "sim_uncoded_bit_errors")
same = all((np.asarray(py[k]) == np.asarray(cc[k])).all() for k in keys)
print(f"python driver (one process): wall {t1 - t0:.2f} s (python + torch import included), frames {[int(x) for x in py['sim_Nframes']]}; counters equal to the C++ run: {same}")
if not same: sys.exit(1)
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- "$EXE" -p "$R/data/params/ber.ini.dvbs2_sweep" -b "$OUT/base" -s 7 -c prof --lanes 1 > "$OUT/run_prof.log" 2> "$OUT/trace.err" || { tail -5 "$OUT/trace.err"; exit 1; }
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/trace" "$OUT/base/codes"
head -12 "$OUT/kernel_stats.csv" | cut -c1-200
