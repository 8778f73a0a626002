#!/usr/bin/env python3
"""Per-kernel means of the SQ counters collected by scripts/pmc_loss_passes.sh:
    python scripts/summarize_pmc.py r02 [--out profiles/r02_loss_passes_pmc.md]
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves; SQ_BUSY_CYCLES per SE; the
ratios below are the interpretable part (MI355X_MICROARCH.md 'rocprofv3 PMC slots')."""
import argparse
import csv
import glob
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]


def load(tag):
    rows = {}
    for f in glob.glob(str(ROOT / "gpurun_out" / tag / "pmc_sq*" / "**" / "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r.get("Kernel_Name", "")
            if "loss_main" not in k:
                continue
            key = (k.split("(")[0], r["Counter_Name"])
            d = rows.setdefault(key, {})
            d[(f, r["Dispatch_Id"])] = d.get((f, r["Dispatch_Id"]), 0.0) + float(r["Counter_Value"])
    out = {}
    for (k, c), d in rows.items():
        v = list(d.values())
        out.setdefault(k, {})[c] = sum(v) / len(v)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    data = load(a.tag)
    lines = ["# SQ counters per launch of the fused-loss passes (config 2, batch 512): scripts/pmc_loss_passes.sh + summarize_pmc.py", ""]
    for k, c in sorted(data.items()):
        lines.append(f"## `{k}`")
        lines.append("")
        lines.append("| counter | mean per launch |")
        lines.append("|---|---|")
        for name in sorted(c):
            lines.append(f"| {name} | {c[name]:.4g} |")
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            def frac(n):
                return f"{c[n] / wc:.3f}" if n in c else "n/a"
            lines.append("")
            lines.append(f"Of the waves' resident cycles: instruction issue active {frac('SQ_ACTIVE_INST_ANY')} "
                         f"(VALU {frac('SQ_ACTIVE_INST_VALU')}, LDS {frac('SQ_ACTIVE_INST_LDS')}), waiting on a counter / barrier "
                         f"{frac('SQ_WAIT_ANY')}, issue-stalled {frac('SQ_WAIT_INST_ANY')} (of which LDS {frac('SQ_WAIT_INST_LDS')}).")
        if "SQ_INSTS_VALU" in c and "SQ_INSTS_MFMA" in c:
            lines.append(f"VALU instructions per MFMA: {c['SQ_INSTS_VALU'] / max(c['SQ_INSTS_MFMA'], 1):.1f}; LDS instructions per MFMA: "
                         f"{c.get('SQ_INSTS_LDS', 0) / max(c['SQ_INSTS_MFMA'], 1):.2f}; LDS bank-conflict cycles / LDS active: "
                         f"{c.get('SQ_LDS_BANK_CONFLICT', 0) / max(c.get('SQ_ACTIVE_INST_LDS', 1), 1):.3f}")
        lines.append("")
    text = "\n".join(lines)
    print(text)
    if a.out:
        pathlib.Path(a.out).write_text(text + "\n")


if __name__ == "__main__":
    main()
