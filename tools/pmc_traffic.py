#!/usr/bin/env python3
"""Turns the FETCH_SIZE / WRITE_SIZE counter CSVs of tools/profile_round.sh into the per-kernel HBM traffic
summary bench.py reads (profiles/*pmc_traffic*.json).

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): both counters are in KiB;
FETCH_SIZE reports half of the bytes actually read (128-byte requests tallied at 64 B) -> x2;
WRITE_SIZE is exact.  Checked here on tests/microbench/rows.hip (profiles/r01_microbench_rows.log).

Usage: tools/pmc_traffic.py <dir with pmc_FETCH_SIZE.csv, pmc_WRITE_SIZE.csv, bench.json> <out.json> [workload name]"""
import collections, csv, json, pathlib, re, sys

src, out = pathlib.Path(sys.argv[1]), pathlib.Path(sys.argv[2])
acc = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    with open(src / f"pmc_{c}.csv") as fh:
        for row in csv.DictReader(fh):
            if row["Counter_Name"] != c:
                continue
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void ", "")
            if "lutldpc" not in k and "kernel" not in k:
                continue
            a = acc[k][c]
            a[0] += float(row["Counter_Value"]); a[1] += 1
kern = {}
for k, v in sorted(acc.items()):
    n = max(v["FETCH_SIZE"][1], 1)
    kern[k] = {"launches": v["FETCH_SIZE"][1],
               "read_bytes_per_launch": v["FETCH_SIZE"][0] / n * 1024 * 2,
               "write_bytes_per_launch": v["WRITE_SIZE"][0] / max(v["WRITE_SIZE"][1], 1) * 1024}
bench = json.loads((src / "bench.json").read_text().strip().splitlines()[-1])
fused = [v for k, v in kern.items() if "pass_fused_kernel" in k]
resident = [v for k, v in kern.items() if "lutldpc_jit_pass" in k]
res = {
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, --kernel-trace only), bench.py --steps 1 --warmup 1; "
            "FETCH_SIZE KiB x 1024 x 2 (gfx950 reports half of the read bytes), WRITE_SIZE KiB x 1024",
    "kernels": kern,
    "fused_pass_hbm_bytes_per_launch": sum(v["read_bytes_per_launch"] + v["write_bytes_per_launch"] for v in fused) if fused else None,
    "algorithmic": {"fused_pass_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"]},
    "resident_hbm_bytes_per_launch": sum(v["read_bytes_per_launch"] + v["write_bytes_per_launch"] for v in resident) if resident else None,
    "workload": (sys.argv[3] if len(sys.argv) > 3 else "dvbs2"), "batch": bench["config"]["frames_per_gpu_per_step"], "mode": "fixed",
    "build": bench["config"]["kernels"].get("build"), "kernel_sources": bench["config"]["kernels"].get("kernel_sources"),
    "message_bytes": bench["config"]["kernels"]["message_bytes"],
}
out.write_text(json.dumps(res, indent=1))
print(json.dumps({k: res[k] for k in ("fused_pass_hbm_bytes_per_launch", "resident_hbm_bytes_per_launch", "algorithmic")}, indent=1))
