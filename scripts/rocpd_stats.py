#!/usr/bin/env python3
"""Per-kernel summary (calls/step, avg us, us/step) from a rocprofv3 rocpd database or kernel_stats CSV.

    python scripts/rocpd_stats.py <results.db | kernel_stats.csv> --steps 13 [--top 30] [--md]
"""
import argparse
import csv
import sqlite3


def rows_from_db(path):
    cur = sqlite3.connect(path).cursor()
    q = "select name, count(*), sum(end - start) from kernels group by name order by 3 desc"
    try:
        return [(n, c, t) for n, c, t in cur.execute(q)]
    except sqlite3.OperationalError:
        q = "select name, count(*), sum(duration) from kernels group by name order by 3 desc"
        return [(n, c, t) for n, c, t in cur.execute(q)]


def rows_from_csv(path):
    return [(r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])) for r in csv.DictReader(open(path))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--top", type=int, default=30)
    ap.add_argument("--md", action="store_true")
    a = ap.parse_args()
    rows = rows_from_db(a.path) if a.path.endswith(".db") else rows_from_csv(a.path)
    tot = sum(r[2] for r in rows)
    print(f"# total GPU time per step: {tot / a.steps / 1e6:.3f} ms; kernel launches per step: "
          f"{sum(r[1] for r in rows) / a.steps:.1f}")
    if a.md:
        print("\n| kernel | calls/step | us/step | avg us | % |\n|---|---|---|---|---|")
    for n, c, t in rows[: a.top]:
        if a.md:
            print(f"| {n[:110]} | {c / a.steps:.1f} | {t / a.steps / 1e3:.1f} | {t / c / 1e3:.1f} | {100 * t / tot:.2f} |")
        else:
            print(f"{n[:80]:80s} {c / a.steps:6.1f} {t / c / 1e3:8.1f} {t / a.steps / 1e3:8.1f} {100 * t / tot:6.2f}")


if __name__ == "__main__":
    main()
