#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output directory: per-kernel time stats and PMC averages."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def find(sub, pat):
    return sorted(glob.glob(os.path.join(root, sub, "**", pat), recursive=True))


print(f"# profile summary for {root}")
for f in find("trace", "*kernel_stats.csv"):
    print(f"\n## kernel stats ({os.path.relpath(f, root)})")
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:12]:
        name = r.get("Name", "")[:70]
        print(f"{name:70s} calls={r.get('Calls')} total_ns={r.get('TotalDurationNs')} avg_ns={r.get('AverageNs')} pct={r.get('Percentage')}")

for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2"):
    files = find(sub, "*counter_collection.csv")
    if not files:
        continue
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")
                acc[k][r.get("Counter_Name")].append(float(r.get("Counter_Value", 0)))
    print(f"\n## {sub}: average counter value per dispatch")
    for k, d in acc.items():
        if "k_fused" not in k and "k_threshold" not in k and "k_morph" not in k and "k_nlm" not in k:
            continue
        short = k[:60]
        for c, v in sorted(d.items()):
            print(f"{short:60s} {c:26s} n={len(v):4d} avg={sum(v)/len(v):.6g}")
