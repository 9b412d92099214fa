#!/bin/bash
# The evidence set of one round for EVERY BASELINE configuration (run on the GPU box through gpurun): per workload the bench line,
# rocprofv3 --kernel-trace --stats of the same command, and the counter passes (HBM traffic; VALU / LDS limiter) summarised by
# tools/pmc_traffic.py / tools/pmc_limiter.py for the workload's dominant kernel.
# Usage: tools/profile_all_configs.sh <out-dir> [workload ...]      then copy <out-dir>/<wl>/{bench.json,kernel_stats.csv,pmc_*.json} to profiles/
set -e
OUT=$1; shift
WLS=${@:-"dvbs2 c2 c5 c5chk c1 twin"}
R=$GRAFT_REPO_ROOT
mkdir -p "$R/$OUT"
for wl in $WLS; do
    args="--workload $wl"
    [ "$wl" = "c2" ] && args="$args --batch 4096"
    kern="lutldpc_jit_pass"
    [ "$wl" = "dvbs2" ] || [ "$wl" = "twin" ] && kern="pass_fused_kernel"
    echo "== $wl ($args; dominant kernel $kern) $(date +%T)"
    BENCH_ARGS="$args" bash "$R/tools/profile_round.sh" "$OUT/$wl" $args --no-configs > "$R/$OUT/$wl.log" 2>&1 || { echo "profile_round failed for $wl"; tail -5 "$R/$OUT/$wl.log"; continue; }
    python3 "$R/tools/pmc_limiter.py" "$R/$OUT/$wl" "$R/$OUT/$wl/pmc_limiter.json" "$kern" "$wl" > /dev/null || echo "limiter summary failed for $wl"
    python3 "$R/tools/pmc_traffic.py" "$R/$OUT/$wl" "$R/$OUT/$wl/pmc_traffic.json" "$wl" > /dev/null || echo "traffic summary failed for $wl"
    python3 - <<PY
import json
b = json.loads(open("$R/$OUT/$wl/bench.json").read().strip().splitlines()[-1])
l = json.load(open("$R/$OUT/$wl/pmc_limiter.json")); t = json.load(open("$R/$OUT/$wl/pmc_traffic.json"))
print("   value %.0f cw/s  frac %.3f  avg launch %.3f ms | valu %.2f lds %.2f conflicts %.2f | traffic fused %s resident %s" % (
    b["value"], b["roofline"]["frac"], b["roofline"]["avg_launch_ms"], l["valu_busy"], l["lds_busy"], l["lds_bank_conflict_share_of_lds_cycles"],
    t.get("fused_pass_hbm_bytes_per_launch"), t.get("resident_hbm_bytes_per_launch")))
PY
done
