#!/usr/bin/env python3
"""What limits the dominant kernel ON THE CHIP, from the SQ counters of tools/profile_round.sh (pmc_sq_a.csv, pmc_sq_b.csv):
per launch of pass_fused_kernel the share of the launch during which the vector ALUs issue, and during which the LDS is busy.

rocprofv3 sums every counter over the 8 XCDs of an MI355X; GRBM_GUI_ACTIVE is therefore 8 x the cycles of the launch.
    valu_busy = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * launch cycles)       (rocprof's VALUBusy)
    lds_busy  = SQ_LDS_IDX_ACTIVE / (256 CUs * launch cycles)                (bank-conflict cycles included)
Usage: tools/pmc_limiter.py <dir> <out.json> [kernel substring] [workload name]"""
import collections, csv, json, pathlib, re, sys

src, out = pathlib.Path(sys.argv[1]), pathlib.Path(sys.argv[2])
want = sys.argv[3] if len(sys.argv) > 3 else "pass_fused_kernel"
acc = collections.defaultdict(lambda: [0.0, 0])
for f in ("pmc_sq_a.csv", "pmc_sq_b.csv"):
    with open(src / f) as fh:
        for row in csv.DictReader(fh):
            if want not in row["Kernel_Name"]:
                continue
            a = acc[(f, row["Counter_Name"])]
            a[0] += float(row["Counter_Value"]); a[1] += 1
m = {}
for (f, c), (s, n) in acc.items():
    m.setdefault(c, s / n)            # GRBM_GUI_ACTIVE is in both passes: keep the first
launch_cycles = m["GRBM_GUI_ACTIVE"] / 8.0
res = {
    "note": "rocprofv3 --pmc (two passes, --kernel-trace only), bench.py --steps 1 --warmup 1, means per launch of " + want,
    "kernel": want, "launches": acc[("pmc_sq_a.csv", "GRBM_GUI_ACTIVE")][1], "counters_per_launch": m,
    "launch_cycles": launch_cycles,
    "valu_busy": m["SQ_ACTIVE_INST_VALU"] * 4 / 1024 / launch_cycles,
    "lds_busy": m["SQ_LDS_IDX_ACTIVE"] / 256 / launch_cycles,
    "lds_bank_conflict_share_of_lds_cycles": m["SQ_LDS_BANK_CONFLICT"] / max(m["SQ_LDS_IDX_ACTIVE"], 1.0),
    "wave_cycles_waiting_share": m["SQ_WAIT_INST_ANY"] / max(m["SQ_WAVE_CYCLES"], 1.0),
    "valu_instructions_per_launch": m["SQ_INSTS_VALU"],
}
bench = src / "bench.json"
if bench.exists():
    b = json.loads(bench.read_text().strip().splitlines()[-1])
    res.update({"workload": (sys.argv[4] if len(sys.argv) > 4 else "dvbs2"), "batch": b["config"]["frames_per_gpu_per_step"], "mode": "fixed",
                "build": b["config"]["kernels"].get("build"), "kernel_sources": b["config"]["kernels"].get("kernel_sources")})
out.write_text(json.dumps(res, indent=1))
print(json.dumps({k: v for k, v in res.items() if k != "counters_per_launch"}, indent=1))
