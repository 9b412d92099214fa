#!/usr/bin/env python3
"""Aggregates rocprofv3 counter_collection.csv files per kernel: mean counter value per dispatch.
Usage: tools/pmc_summary.py <dir> [name-filter]"""
import csv, sys, collections, pathlib, re

root = pathlib.Path(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in root.rglob("*counter_collection.csv"):
    with open(f) as fh:
        for row in csv.DictReader(fh):
            k = re.sub(r"\(.*", "", row["Kernel_Name"]).replace("void lutldpc::", "")
            if flt and flt not in k:
                continue
            a = acc[k][row["Counter_Name"]]
            a[0] += float(row["Counter_Value"]); a[1] += 1
for k in sorted(acc):
    print(k)
    for c in sorted(acc[k]):
        s, n = acc[k][c]
        print(f"    {c:28s} {s / n:16.1f}   (n={n})")
