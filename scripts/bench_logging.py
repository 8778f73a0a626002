#!/usr/bin/env python3
"""Isolated timing of the two passes of the fused sampled loss at the bench shape (nothing else on the GPU):
    python scripts/bench_logging.py [--batch 512] [--hidden 128] [--items 3883] [--reps 20]
HIP events around the kernel through xfmr_loss_cfg.profile_grad / profile_log. XFMR_HIP_LIB selects another build for A/B runs."""
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
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--head", default="InfoNCELoss")
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
    lib = N.load()
    print("library:", N.LIB_PATH)
    n_cols = int(torch.unique(neg).numel())
    a, b = ctypes.c_void_p(), ctypes.c_void_p()
    N.check(lib.xfmr_event_create(ctypes.byref(a), 1), "xfmr_event_create")
    N.check(lib.xfmr_event_create(ctypes.byref(b), 1), "xfmr_event_create")
    pair = (a.value, b.value)
    for name, kw, which, flops in (
        ("gradient pass", dict(all_heads=False, need_grad=True), "profile_grad", 4.0 * B * L * n_cols * H),
        ("logging pass, all 7 heads (-1)", dict(all_heads=True, need_grad=False), "profile_log", 2.0 * B * L * n_cols * H),
        ("logging pass without the InfoNCE lse (-2)", dict(all_heads=2, need_grad=False), "profile_log", 2.0 * B * L * n_cols * H),
    ):
        ms = []
        for _ in range(args.reps + 3):
            ops.sampled_loss(tok, mask, pos, neg, table, rn, train_head=args.head, precision="bf16", table_bf16=tb,
                             **kw, **{which: pair})
            torch.cuda.synchronize()
            t = ctypes.c_float()
            if lib.xfmr_event_elapsed_ms(pair[0], pair[1], ctypes.byref(t)) == 0:
                ms.append(t.value)
        ms = sorted(ms[3:])
        med = ms[len(ms) // 2]
        print(f"{name}: median {med * 1e3:.1f} us (min {ms[0] * 1e3:.1f}) -> {flops / (med * 1e-3) / 1e12:.0f} TFLOP/s executed")


if __name__ == "__main__":
    main()
