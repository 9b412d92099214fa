#!/bin/bash
# Run-to-run levels: N processes of the same short bench under rocprofv3 with the instruction-cache counters of the SQC
# (names taken from the counter list of the box) + wave cycles; per process the mean duration of the fused kernel and the
# counter sums go to <out>/summary.txt.  Usage: tools/levels_probe2.sh <out-dir> <N>
OUT=$1; N=${2:-6}
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > "$R/$OUT/counters_all.txt" 2>&1
grep -o -E "\b(SQC_ICACHE_[A-Z_]+|SQ_IFETCH[A-Z_]*|SQC_INST_[A-Z_]+|SQ_INST_LEVEL_[A-Z_]+|SQC_DCACHE_[A-Z_]+)\b" "$R/$OUT/counters_all.txt" | sort -u > "$R/$OUT/counters_sqc.txt"
CTR=$(grep -E "^SQC_ICACHE_(REQ|HITS|MISSES|MISSES_DUPLICATE)$" "$R/$OUT/counters_sqc.txt" | tr '\n' ' ')
echo "counters: $CTR" > "$R/$OUT/summary.txt"
for i in $(seq 1 "$N"); do
    rocprofv3 --pmc $CTR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$R/$OUT/run$i" -- python3 "$R/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-kernel-events --frame-loop-steps 0 --as-shipped-steps 0 --reps 0 > "$R/$OUT/run$i.json" 2> "$R/$OUT/run$i.err"
    python3 - "$R/$OUT/run$i" >> "$R/$OUT/summary.txt" <<'PY'
import csv, glob, sys, collections
d = sys.argv[1]
dur = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pass_fused" in r.get("Kernel_Name", ""):
            dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
acc = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "pass_fused" in r.get("Kernel_Name", ""):
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
n = max(len(dur), 1)
print(d.split("/")[-1], "fused launches", len(dur), "mean_us %.1f" % (sum(dur) / n), " ".join(f"{k}={v / n:.4g}" for k, v in sorted(acc.items())))
PY
    tail -1 "$R/$OUT/summary.txt"
    rm -rf "$R/$OUT/run$i"          # the raw csv files are large
done
