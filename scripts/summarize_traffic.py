#!/usr/bin/env python3
"""Memory-side traffic per kernel of the training step from scripts/pmc_traffic.sh:
    python scripts/summarize_traffic.py r02 > profiles/r02_step_traffic.md
FETCH_SIZE is doubled (gfx950 reports half the bytes of 16-byte-per-lane reads: MI355X_MICROARCH.md, HBM section); WRITE_SIZE
is taken as reported. Both count requests at the L2's memory side -- Infinity Cache hits included."""
import collections
import csv
import pathlib
import sys

tag = sys.argv[1]
root = pathlib.Path(__file__).resolve().parents[1] / "gpurun_out" / tag


def agg(sub):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(root / sub / "step_counter_collection.csv")):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return d


f, w, l2 = agg("pmc_fetch"), agg("pmc_write"), agg("pmc_l2")
dur = collections.defaultdict(list)
for r in csv.DictReader(open(root / "pmc_fetch" / "step_kernel_trace.csv")):
    dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print(f"# Memory-side traffic per kernel of the training step (config 2, batch 512): scripts/pmc_traffic.sh {tag} + summarize_traffic.py\n")
print("FETCH_SIZE doubled (gfx950 correction), WRITE_SIZE as reported, MB per launch; L2 hit rate = TCC_HIT / (TCC_HIT + TCC_MISS);")
print("TB/s = (read + written) / the launch's duration in the FETCH_SIZE pass (counter collection slows launches by a few %).\n")
print("| kernel | launches | read MB | written MB | TB/s | L2 hit rate |\n|---|---|---|---|---|---|")
rows = []
for k in f:
    fs = f[k]["FETCH_SIZE"]
    ws = w.get(k, {}).get("WRITE_SIZE", [0.0])
    h, m = sum(l2.get(k, {}).get("TCC_HIT_sum", [0])), sum(l2.get(k, {}).get("TCC_MISS_sum", [0]))
    rd, wr = 2 * sum(fs) / len(fs) / 1e3, sum(ws) / len(ws) / 1e3
    t = sum(dur[k]) / len(dur[k]) if dur.get(k) else 0
    rows.append((rd + wr, k, len(fs), rd, wr, (rd + wr) * 1e6 / t / 1e3 if t else 0, h / max(1.0, h + m)))
for tot, k, n, rd, wr, tb, hit in sorted(rows, reverse=True):
    if tot < 0.5:
        continue
    name = k.replace("(anonymous namespace)::", "").split("(")[0]
    print(f"| `{name[:70]}` | {n} | {rd:.1f} | {wr:.1f} | {tb:.2f} | {hit:.2f} |")
