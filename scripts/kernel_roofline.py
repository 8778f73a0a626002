#!/usr/bin/env python3
"""Per-kernel achieved rates of the default bench line (config 2, batch 512) from a rocprofv3 kernel_stats CSV.

    python scripts/kernel_roofline.py profiles/r01_kernel_stats.csv > profiles/r01_kernel_roofline.md

Algorithmic bytes = the tensors a launch must read and write once (DESIGN.md section 3 says which are bf16);
re-reads that hit L2 / Infinity Cache are not counted. HBM peak 8 TB/s nominal (MI355X_MICROARCH.md); a plain
streaming kernel on tensors of these sizes measures 5-6.6 TB/s (scripts/bw_probe.py, scripts/probe/tile_copy.hip).
MFMA peak 2.5 PFLOP/s dense bf16.
"""
import csv
import sys

B, L, H, I, A, NL, ND = 512, 200, 128, 512, 4, 4, 3883
T = B * L
MB = 1e6
f32, b16 = 4, 2
TH, TI, T3H = T * H, T * I, T * 3 * H

# kernel-name fragment -> (description, algorithmic bytes per launch (average over the launches that share the name), flops)
K = [
    ("gemm_kernel<PrecBF16, 64, 128, 64, false, false, 0,", "QKV Linear fwd", TH * b16 + T3H * b16, 2 * T * 3 * H * H),
    ("ffn_fwd_fused_kernel", "FFN fwd in one kernel: FFN1 + GELU + FFN2 + dropout + residual + LayerNorm (writes u, g)",
     TH * b16 + TH * f32 + 2 * TI * b16 + 2 * TH * f32 + TH * b16, 4 * T * I * H),
    ("gemm_kernel<PrecBF16, 64, 128, 64, false, false, 1,", "FFN1 Linear fwd + GELU (+ gelu') [two-kernel form]", TH * b16 + 2 * TI * b16, 2 * T * I * H),
    ("gemm_kernel<PrecBF16, 64, 128, 64, false, false, 5,", "out-proj Linear fwd + dropout + residual + LayerNorm",
     TH * b16 + TH * f32 * 3 + TH * b16, 2 * T * H * H),
    ("ffn_bwd_dx_fused_kernel", "FFN bwd dX chain in one kernel: FFN2 dX x gelu'(u) -> dI -> FFN1 dX + LayerNorm bwd",
     TH * b16 + 2 * TI * b16 + TH * f32 * 3 + TH * b16, 4 * T * I * H),
    ("gemm_kernel<PrecBF16, 64, 128, 64, false, true, 6,", "QKV dX + residual grad + LayerNorm bwd",
     T3H * b16 + TH * f32 * 3 + TH * b16, 2 * T * H * 3 * H),
    ("gemm_kernel<PrecBF16, 64, 64, 64, false, true, 3,", "FFN2 dX x gelu' [two-kernel form]", TH * b16 + 2 * TI * b16, 2 * T * I * H),
    ("gemm_kernel<PrecBF16, 64, 64, 64, false, true, 0, 7u", "out-proj dX", 2 * TH * b16, 2 * T * H * H),
    ("gemm_kernel<PrecBF16, 64, 64, 64, false, true, 0, 3u", "QKV dX + residual grad (layer 0)", T3H * b16 + 2 * TH * f32, 2 * T * H * 3 * H),
    ("gemm_kernel<PrecBF16, 128, 64, 128, true, true, 4,", "dW split-K (avg of the 4 weights; operands only, + 14 MB of slabs)",
     ((TH + TI) * b16 * 2 + (TH + TH) * b16 + (T3H + TH) * b16) / 4, (2 * 2 * T * H * I + 2 * T * H * H + 2 * T * 3 * H * H) / 4),
    ("dw_ring_kernel", "dW in token slabs (LDS-DMA ring), the four weights of a layer in one launch (in-line form; operands only, + 38 MB of slabs)",
     (TH + TI) * b16 * 2 + (TH + TH) * b16 + (T3H + TH) * b16, 2 * 2 * T * H * I + 2 * T * H * H + 2 * T * 3 * H * H),
    ("attn_fwd_seq_bf16_kernel", "attention fwd (per layer)", T3H * b16 + TH * b16 + B * A * L * f32, 4 * B * A * L * (L + 1) / 2 * 32),
    ("attn_bwd_fused_bf16_kernel", "attention bwd (per layer)", 2 * T3H * b16 + 2 * TH * b16, 10 * B * A * L * (L + 1) / 2 * 32),
    ("ln_bwd_v4_kernel", "LayerNorm bwd (top layer / embedding)", TH * f32 * 3 + TH * b16 / 2, 0),
    ("ln_fwd_v4_kernel<32, true>", "embedding gather + LayerNorm", TH * f32 * 3 + TH * b16, 0),
    ("loss_main_dma_kernel<128, 7>", "loss gradient pass", 0, 4.0 * T * ND * H),
    ("loss_main_dma_kernel<128, -3>", "loss logging pass (6 heads + statistics; masked fast path)", 0, 2.0 * T * ND * H),
    ("loss_main_dma_kernel<128, -2>", "loss logging pass (6 heads + statistics)", 0, 2.0 * T * ND * H),
    ("loss_combine_kernel", "loss combine (values only)", 0, 0),
    ("multi_rowsum_kernel", "split-K slab / partial-record reduction", 247e6, 0),
    ("scale_kernel", "d_tok *= upstream gradient (returns at once when it is 1)", 0, 0),
]

import argparse
import json

ap = argparse.ArgumentParser()
ap.add_argument("csv")
ap.add_argument("--steps", type=int, default=13)
ap.add_argument("--json", action="store_true", help="print the GEMM-family summary as JSON instead of the table")
args = ap.parse_args()
rows = list(csv.DictReader(open(args.csv)))
steps = args.steps
lines = [f"# Achieved rates per kernel of the default bench line (config 2, batch 512): python scripts/kernel_roofline.py {args.csv}", "",
         "| kernel | launches/step | avg µs | algorithmic MB | TB/s (of 8 nominal; 5–6.6 measured stream) | TFLOP/s (of 2500) |",
         "|---|---|---|---|---|---|"]
gemm_bytes = gemm_ns = 0.0
for frag, desc, nbytes, flops in K:
    hit = [r for r in rows if frag in r["Name"]]
    if not hit:
        continue
    calls = sum(int(r["Calls"]) for r in hit)
    tot = sum(float(r["TotalDurationNs"]) for r in hit)
    avg = tot / calls / 1e3
    tb = f"{nbytes / (avg * 1e-6) / 1e12:.2f}" if nbytes else "—"
    tf = f"{flops / (avg * 1e-6) / 1e12:.0f}" if flops else "—"
    lines.append(f"| {desc} (`{frag.strip(', ')}`) | {calls / steps:.1f} | {avg:.1f} | {nbytes / MB:.0f} | {tb} | {tf} |")
    if frag.startswith(("gemm_kernel", "gemm_group_kernel", "dw_ring_kernel", "ffn_")):
        gemm_bytes += nbytes * calls / steps
        gemm_ns += tot / steps
if args.json:
    print(json.dumps({"gemm_family_tbps": round(gemm_bytes / (gemm_ns * 1e-9) / 1e12, 3),
                      "gemm_family_ms_per_step": round(gemm_ns / 1e6, 4),
                      "gemm_family_compulsory_gb_per_step": round(gemm_bytes / 1e9, 3)}))
else:
    lines.append("")
    lines.append(f"GEMM family: {gemm_bytes / 1e9:.2f} GB of compulsory bytes in {gemm_ns / 1e6:.3f} ms per step = "
                 f"{gemm_bytes / (gemm_ns * 1e-9) / 1e12:.2f} TB/s")
    print("\n".join(lines))
