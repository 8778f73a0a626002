#!/usr/bin/env python3
"""Per-kernel SQ counter ratios of the whole step (scripts/pmc_step.sh): python scripts/summarize_pmc_step.py r02 [--out f.md]"""
import argparse, csv, glob, pathlib, re
ROOT = pathlib.Path(__file__).resolve().parents[1]
ap = argparse.ArgumentParser(); ap.add_argument("tag"); ap.add_argument("--out", default=None); a = ap.parse_args()
agg = {}
for f in glob.glob(str(ROOT / "gpurun_out" / a.tag / "pmc_step*" / "**" / "*counter_collection.csv"), recursive=True):
    per = {}
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0][:80]
        per.setdefault((k, r["Dispatch_Id"]), {}).setdefault(r["Counter_Name"], 0.0)
        per[(k, r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    for (k, _), c in per.items():
        d = agg.setdefault(k, {"n": {}, "s": {}})
        for name, v in c.items():
            d["s"][name] = d["s"].get(name, 0.0) + v
            d["n"][name] = d["n"].get(name, 0) + 1
rows = []
for k, d in agg.items():
    m = {n: d["s"][n] / d["n"][n] for n in d["s"]}
    wc = m.get("SQ_WAVE_CYCLES", 0)
    if not wc: continue
    rows.append((d["s"]["SQ_WAVE_CYCLES"], k, m, d["n"]["SQ_WAVE_CYCLES"]))
rows.sort(reverse=True)
lines = ["# SQ counters per kernel of the training step (config 2, batch 512): scripts/pmc_step.sh + summarize_pmc_step.py",
         "", "Shares of the waves' resident cycles (quad-cycle counters): issue = SQ_ACTIVE_INST_ANY, of which VALU / LDS; wait = SQ_WAIT_ANY",
         "(s_waitcnt / barrier); stall = SQ_WAIT_INST_ANY. VALU/MFMA/VMEM/LDS = wave-instructions per launch.", "",
         "| kernel | launches | issue | VALU | LDS | wait | stall | LDS conflict / LDS active | VALU insts | MFMA | VMEM | LDS insts |", "|---|---|---|---|---|---|---|---|---|---|---|---|"]
for _, k, m, n in rows[:24]:
    wc = m["SQ_WAVE_CYCLES"]
    f = lambda x: f"{m.get(x, 0) / wc:.2f}"
    lc = m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_ACTIVE_INST_LDS", 1), 1)
    g = lambda x: f"{m.get(x, 0):.3g}"
    lines.append(f"| `{k}` | {n} | {f('SQ_ACTIVE_INST_ANY')} | {f('SQ_ACTIVE_INST_VALU')} | {f('SQ_ACTIVE_INST_LDS')} | {f('SQ_WAIT_ANY')} | {f('SQ_WAIT_INST_ANY')} | {lc:.2f} | {g('SQ_INSTS_VALU')} | {g('SQ_INSTS_MFMA')} | {g('SQ_INSTS_VMEM')} | {g('SQ_INSTS_LDS')} |")
text = "\n".join(lines)
print(text)
if a.out: pathlib.Path(a.out).write_text(text + "\n")
