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


def steps_in(stats_csv, default):
    """Training steps in a kernel_stats CSV = launches of the AdamW kernel (one per step)."""
    for r in csv.DictReader(open(stats_csv)):
        if "adamw_kernel" in r["Name"]:
            return int(r["Calls"])
    return default


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--steps", type=int, default=0, help="steps in the traces (default: counted from the AdamW launches)")
    a = ap.parse_args()
    src = ROOT / "gpurun_out" / a.tag
    prof = ROOT / "profiles"
    line = [l for l in open(src / "bench.json") if l.startswith("{")][-1]
    (prof / f"{a.tag}_bench_line.json").write_text(json.dumps(json.loads(line), indent=1) + "\n")
    for sub, suffix in (("trace", ""), ("trace_overlap", "_overlap")):
        stats = find(src / sub / "**" / "*kernel_stats.csv")
        if not stats:
            continue
        n_steps = a.steps or steps_in(stats, 48)
        if not suffix:
            shutil.copy(stats, prof / f"{a.tag}_kernel_stats.csv")
            main_steps = n_steps
        md = subprocess.run([sys.executable, str(ROOT / "scripts" / "rocpd_stats.py"), stats, "--steps", str(n_steps),
                             "--top", "40", "--md"], capture_output=True, text=True, check=True).stdout
        head = (f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline"
                f"{' --overlap on' if suffix else ' --overlap off'}  ({a.tag}; {n_steps} steps profiled)\n")
        (prof / f"{a.tag}_kernel_stats{suffix}.md").write_text(head + md)
    for v, what in (("config4", "--preset config4"), ("config5", "--preset config5"), ("b32", "--batch 32")):
        stats = find(src / f"trace_{v}" / "**" / "*kernel_stats.csv")
        if not stats:
            continue
        n_steps = a.steps or steps_in(stats, 48)
        md = subprocess.run([sys.executable, str(ROOT / "scripts" / "rocpd_stats.py"), stats, "--steps", str(n_steps),
                             "--top", "30", "--md"], capture_output=True, text=True, check=True).stdout
        head = (f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --spinup-steps 0 --no-cpu-baseline "
                f"--overlap off --graph off {what}  ({a.tag}; {n_steps} steps profiled)\n")
        (prof / f"{a.tag}_{v}_kernel_stats.md").write_text(head + md)
    f = find(src / "pmc_FETCH_SIZE" / "**" / "*counter_collection.csv")
    w = find(src / "pmc_WRITE_SIZE" / "**" / "*counter_collection.csv")
    static = {"source": f"profiles/{a.tag}_* (rocprofv3 runs of scripts/collect_profiles.sh {a.tag}; static, not measured in the bench run)",
              "round": a.tag}
    traffic = {}
    if f and w:
        for label, needle in (("gradient_pass", "loss_main_dma_kernel<128, 7>"), ("logging_pass", "loss_main_dma_kernel<128, -3>")):
            fk, wk = pmc_mean(f, "FETCH_SIZE", needle), pmc_mean(w, "WRITE_SIZE", needle)
            if fk is not None and wk is not None:
                traffic[label] = {"kernel": needle, "FETCH_SIZE_KB": fk, "WRITE_SIZE_KB": wk,
                                  "hbm_bytes_per_launch": (2 * fk + wk) * 1024}
        rec = {"batch": 512, "precision": "bf16", "kernels": traffic,
               "correction": "gfx950: FETCH_SIZE reports half of the bytes of wide (16 B/lane) reads -> doubled; "
                             "WRITE_SIZE exact (MI355X_MICROARCH.md, HBM); separate --pmc passes",
               "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- python3 scripts/bench_logging.py --reps 4",
               "round": a.tag}
        (prof / f"{a.tag}_loss_traffic.json").write_text(json.dumps(rec, indent=1) + "\n")
        if "logging_pass" in traffic:
            static["dominant_kernel_hbm_bytes_per_launch"] = traffic["logging_pass"]["hbm_bytes_per_launch"]
    # per-kernel HBM traffic of the whole step (scripts/pmc_traffic.sh): bytes per launch, FETCH_SIZE doubled + WRITE_SIZE
    sf = find(src / "pmc_fetch" / "**" / "*counter_collection.csv")
    sw = find(src / "pmc_write" / "**" / "*counter_collection.csv")
    if sf and sw:
        per = {}
        for key, needle in (("ffn_fwd_fused_kernel", "ffn_fwd_fused_kernel"), ("ffn_bwd_dx_fused_kernel", "ffn_bwd_dx_fused_kernel"),
                            ("attn_fwd_seq_bf16_kernel", "attn_fwd_seq_bf16_kernel"),
                            ("attn_bwd_fused_bf16_kernel", "attn_bwd_fused_bf16_kernel"),
                            ("loss_logging_pass", "loss_main_dma_kernel<128, -3>"),
                            ("loss_gradient_pass", "loss_main_dma_kernel<128, 7>"),
                            ("dw_ring_kernel", "dw_ring_kernel"), ("multi_rowsum_kernel", "multi_rowsum_kernel")):
            fk, wk = pmc_mean(sf, "FETCH_SIZE", needle), pmc_mean(sw, "WRITE_SIZE", needle)
            if fk is not None and wk is not None:
                per[key] = (2 * fk + wk) * 1024
        if per:
            static["kernel_hbm_bytes_per_launch"] = per
            static["kernel_hbm_bytes_source"] = (f"profiles/{a.tag}_step_traffic.md (scripts/pmc_traffic.sh {a.tag}: rocprofv3 --pmc "
                                                 "FETCH_SIZE | WRITE_SIZE over two bench steps, one stream; FETCH_SIZE doubled)")
        md = subprocess.run([sys.executable, str(ROOT / "scripts" / "summarize_traffic.py"), a.tag], capture_output=True, text=True)
        if md.returncode == 0 and md.stdout:
            (prof / f"{a.tag}_step_traffic.md").write_text(md.stdout)
    # GEMM family: compulsory bytes / GPU time of every gemm_kernel launch of one step (scripts/kernel_roofline.py table)
    stats = prof / f"{a.tag}_kernel_stats.csv"
    if stats.exists():
        a.steps = a.steps or steps_in(stats, 48)
        out = subprocess.run([sys.executable, str(ROOT / "scripts" / "kernel_roofline.py"), str(stats), "--steps", str(a.steps),
                              "--json"], capture_output=True, text=True, check=True).stdout
        fam = json.loads(out)
        static |= fam
        md = subprocess.run([sys.executable, str(ROOT / "scripts" / "kernel_roofline.py"), str(stats), "--steps", str(a.steps)],
                            capture_output=True, text=True, check=True).stdout
        (prof / f"{a.tag}_kernel_roofline.md").write_text(md)
    # vector-instruction counts of the dominant kernel (the masked logging pass is bound by VALU issue, not by the matrix core)
    sys.path.insert(0, str(ROOT / "scripts"))
    import summarize_pmc

    for k, c in summarize_pmc.load(a.tag).items():
        if "loss_main_dma_kernel<128, -3>" in k and "SQ_INSTS_VALU" in c:
            static["dominant_kernel_valu"] = {
                "kernel": k, "wave_instructions_per_launch": c["SQ_INSTS_VALU"],
                "active_quad_cycles_per_launch": c.get("SQ_ACTIVE_INST_VALU"),
                "mfma_wave_instructions_per_launch": c.get("SQ_INSTS_MFMA"),
                "source": f"profiles/{a.tag}_loss_passes_pmc.md (rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU ..., scripts/pmc_loss_passes.sh)"}
    (prof / f"{a.tag}_bench_static.json").write_text(json.dumps(static, indent=1) + "\n")
    pm = subprocess.run([sys.executable, str(ROOT / "scripts" / "summarize_pmc.py"), a.tag, "--out",
                         str(prof / f"{a.tag}_loss_passes_pmc.md")], capture_output=True, text=True)
    print(pm.stdout[-400:])
    print("profiles/ updated from", src)


if __name__ == "__main__":
    main()
