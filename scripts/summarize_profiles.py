#!/usr/bin/env python3
"""Turn gpurun_out/<tag>/ (scripts/collect_profiles.sh) into the tracked files under profiles/.

    python scripts/summarize_profiles.py r01 --steps 13

Writes profiles/<tag>_bench_line.json, profiles/<tag>_kernel_stats.md (+ _overlap), profiles/<tag>_kernel_stats.csv
and profiles/loss_main_traffic.json (HBM bytes per launch of the gradient-pass loss kernel, FETCH_SIZE doubled on
gfx950 as MI355X_MICROARCH.md prescribes for 16-byte-per-lane reads).
"""
import argparse
import csv
import glob
import json
import pathlib
import shutil
import subprocess
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]


def find(pattern):
    hits = glob.glob(str(pattern), recursive=True)
    return hits[0] if hits else None


def pmc_mean(path, counter, needle):
    vals = {}
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter or needle not in r.get("Kernel_Name", ""):
            continue
        vals.setdefault(r["Dispatch_Id"], 0.0)
        vals[r["Dispatch_Id"]] += float(r["Counter_Value"])
    v = list(vals.values())
    return sum(v) / len(v) if v else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--steps", type=int, default=13)
    ap.add_argument("--kernel", default="loss_main_dma_kernel<128, 7>")
    a = ap.parse_args()
    src = ROOT / "gpurun_out" / a.tag
    prof = ROOT / "profiles"
    line = [l for l in open(src / "bench.json") if l.startswith("{")][-1]
    (prof / f"{a.tag}_bench_line.json").write_text(json.dumps(json.loads(line), indent=1) + "\n")
    for sub, suffix in (("trace", ""), ("trace_overlap", "_overlap")):
        stats = find(src / sub / "**" / "*kernel_stats.csv")
        if not stats:
            continue
        if not suffix:
            shutil.copy(stats, prof / f"{a.tag}_kernel_stats.csv")
        md = subprocess.run([sys.executable, str(ROOT / "scripts" / "rocpd_stats.py"), stats, "--steps", str(a.steps),
                             "--top", "40", "--md"], capture_output=True, text=True, check=True).stdout
        head = (f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline"
                f"{' --overlap on' if suffix else ' --overlap off'}  ({a.tag}; {a.steps} steps profiled)\n")
        (prof / f"{a.tag}_kernel_stats{suffix}.md").write_text(head + md)
    f = find(src / "pmc_FETCH_SIZE" / "**" / "*counter_collection.csv")
    w = find(src / "pmc_WRITE_SIZE" / "**" / "*counter_collection.csv")
    if f and w:
        fk = pmc_mean(f, "FETCH_SIZE", a.kernel)
        wk = pmc_mean(w, "WRITE_SIZE", a.kernel)
        if fk is not None and wk is not None:
            rec = {
                "kernel": a.kernel + " (gradient pass of the fused sampled loss)", "batch": 512, "precision": "bf16",
                "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk, "hbm_bytes_per_launch": (2 * fk + wk) * 1024,
                "correction": "gfx950: FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads -> doubled; "
                              "WRITE_SIZE exact (MI355X_MICROARCH.md, HBM); separate --pmc passes",
                "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 scripts/bench_loss.py --reps 4",
                "round": a.tag,
            }
            mf = find(src / "pmc_MFMA" / "**" / "*counter_collection.csv")
            if mf:
                busy = pmc_mean(mf, "SQ_VALU_MFMA_BUSY_CYCLES", a.kernel)
                active = pmc_mean(mf, "GRBM_GUI_ACTIVE", a.kernel)
                if busy and active:
                    # SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_32x32x16_bf16 (MI355X_MICROARCH.md), summed
                    # over the 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs (8 x the launch's cycles)
                    rec |= {"SQ_VALU_MFMA_BUSY_CYCLES": busy, "GRBM_GUI_ACTIVE": active,
                            "mfma_instructions_per_launch": busy / 32.0,
                            "mfma_busy_fraction": busy / (1024.0 * active / 8.0),
                            "mfma_note": "matrix-core busy cycles summed over 1024 SIMDs / (1024 x active cycles of "
                                         "the launch, GRBM_GUI_ACTIVE / 8 XCDs); its own --pmc pass"}
            (prof / "loss_main_traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
            print(rec)
    print("profiles/ updated from", src)


if __name__ == "__main__":
    main()
