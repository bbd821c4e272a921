#!/usr/bin/env python3
"""Prints a compact per-kernel table from a rocprofv3 --kernel-trace --stats csv directory."""
import csv, glob, sys
d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 20
f = glob.glob(d + '/**/*_kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:top]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:64]:64s} n={r['Calls']:>6s} avg={float(r['AverageNs'])/1e3:8.2f}us min={float(r['MinNs'])/1e3:7.2f} tot={float(r['TotalDurationNs'])/1e6:8.2f}ms {float(r['Percentage']):5.1f}%")
print(f"total {tot/1e6:.2f} ms")
