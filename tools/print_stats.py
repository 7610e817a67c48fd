#!/usr/bin/env python3
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    i = n.find("k_")
    short = n[i:i + 26] if i >= 0 else n[:26]
    print(f"{short:28s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs'])/1e3:10.1f} pct={r['Percentage']}")
