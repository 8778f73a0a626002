#!/usr/bin/env python3
"""Micro-benchmark of the fused sampled-loss launch sequence at the MovieLens-1M bench shape.

    python scripts/bench_loss.py [--batch 512] [--seq-len 200] [--hidden 128] [--items 3883] [--reps 10]

Prints, per variant (all heads / train head only; bf16 / fp32 MFMA), the average duration of loss_main_kernel
(HIP events recorded around it on the launch stream) and of the whole launch sequence, with the achieved
TFLOP/s on the executed 4*Np*Nd*H flops (Nd = distinct negative items).
"""

from __future__ import annotations

import argparse
import ctypes
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "transformer-recommenders_amd"):
    sys.path.insert(0, str(p))

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--seq-len", type=int, default=200)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--items", type=int, default=3883)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--head", default="InfoNCELoss")
    ap.add_argument("--fp32", action="store_true")
    args = ap.parse_args()
    from xfmr_rec_amd import _native as N
    from xfmr_rec_amd import ops

    dev = "cuda"
    B, L, H, V = args.batch, args.seq_len, args.hidden, args.items
    g = torch.Generator().manual_seed(0)
    table = torch.randn(V + 1, H, generator=g)
    table = table / table.norm(dim=-1, keepdim=True)
    table[0] = 0
    table = table.to(dev)
    rn, tb = ops.table_prepare(table)
    tok = torch.randn(B * L, H, generator=g).to(dev)
    mask = torch.ones(B * L, dtype=torch.uint8, device=dev)
    pos = torch.randint(1, V + 1, (B * L,), generator=g).to(dev)
    neg = torch.randint(1, V + 1, (B * L,), generator=g).to(dev)
    hip = ctypes.CDLL("libamdhip64.so.7")  # soname: resolves to the HIP runtime torch already loaded
    lib = N.load()
    n_cols = int(torch.unique(neg).numel())  # the kernels walk the distinct negative items
    flops = 4.0 * (B * L) * n_cols * H
    precs = ["bf16"] + (["fp32"] if args.fp32 else [])
    for prec in precs:
        for all_heads in (True, False):
            evs = []
            for _ in range(args.reps + 2):
                a, b = ctypes.c_void_p(), ctypes.c_void_p()
                hip.hipEventCreate(ctypes.byref(a))
                hip.hipEventCreate(ctypes.byref(b))
                evs.append((a, b))
            outer = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in evs]
            for i, (a, b) in enumerate(evs):
                lib.xfmr_sampled_loss_profile_next(a, b)
                outer[i][0].record()
                losses, stats, d = ops.sampled_loss(tok, mask, pos, neg, table, rn, train_head=args.head,
                                                    all_heads=all_heads, precision=prec, table_bf16=tb)
                outer[i][1].record()
            torch.cuda.synchronize()
            ms = []
            for a, b in evs[2:]:
                t = ctypes.c_float()
                hip.hipEventElapsedTime(ctypes.byref(t), a, b)
                ms.append(t.value)
            tot = [x.elapsed_time(y) for x, y in outer[2:]]
            k = sum(ms) / len(ms)
            print(f"{prec:5s} all_heads={all_heads!s:5s} main {k:8.3f} ms  ({flops / k / 1e9:8.1f} TFLOP/s)  "
                  f"sequence {sum(tot) / len(tot):8.3f} ms  loss={losses[N.LOSS_IDS[args.head]].item():.2f}", flush=True)


if __name__ == "__main__":
    main()
