#!/usr/bin/env python3
"""Top kernels of a rocprofv3 kernel_stats CSV: python scripts/prof_top.py gpurun_out/<tag>/p_kernel_stats.csv [n]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    print(f"{r['Name'][:105]:105s} {int(r['Calls']):5d} {float(r['AverageNs'])/1e3:8.1f} us {float(r['Percentage']):5.1f}%")
print(f"total {tot/1e6:.2f} ms")
