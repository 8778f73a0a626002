#!/usr/bin/env python3
"""Phase timeline of ffn_fwd_fused_kernel from s_memtime stamps (probe build: hipcc -DXF_FFN_STAMP, see gemm.hip).
    XFMR_HIP_LIB=build/libxfmr_hip_stamp.so python scripts/probe/ffn_fwd_stamps.py"""
import ctypes as C
import os
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "transformer-recommenders_amd"))
from xfmr_rec_amd import _native as N  # noqa: E402

lib = N.load()
DEV = "cuda"
H, I = 128, 512
NAMES = ["x tile -> LDS -> fragments", "loop entry barrier", "commit W1 + barrier", "GEMM1 + u -> sG", "barrier",
         "pass (u, g stores; gelu) + commit W2", "barrier", "GEMM2 + loop barrier"]


def run(M):
    g = torch.Generator().manual_seed(0)
    x16 = torch.randn(M, H, generator=g).to(DEV).to(torch.bfloat16)
    w1 = (torch.randn(I, H, generator=g) * 0.08).to(DEV).to(torch.bfloat16)
    b1 = torch.zeros(I, device=DEV)
    w2 = (torch.randn(H, I, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
    b2 = torch.zeros(H, device=DEV)
    res = torch.randn(M, H, generator=g).to(DEV)
    gamma, beta = torch.ones(H, device=DEV), torch.zeros(H, device=DEV)
    u = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
    gg = torch.empty_like(u)
    pre, y = torch.empty(M, H, device=DEV), torch.empty(M, H, device=DEV)
    y16 = torch.empty(M, H, device=DEV, dtype=torch.bfloat16)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    nblk = ((M + 63) // 64 + 7) // 8 * 8
    stamps = torch.zeros(nblk, 64, dtype=torch.int64, device=DEV)
    os.environ["XFMR_FFN_STAMPS"] = hex(stamps.data_ptr())
    fn = lib.xf_ffn_fwd_fused_ex
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_float, N.Seed, C.c_uint32,
                                      C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 5
    for _ in range(3):
        rc = fn(N.ptr(x16), N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2), N.ptr(u), N.ptr(gg), N.ptr(pre), M, H, I, N.ptr(res),
                0.1, 5, 9, N.ptr(gamma), N.ptr(beta), 1e-12, N.ptr(y), N.ptr(y16), N.ptr(mean), N.ptr(rstd), N.stream())
        assert rc == 0
    torch.cuda.synchronize()
    s = stamps.cpu().numpy()[: (M + 63) // 64].astype("float64")
    s = s[s[:, 16] > 0]
    t0 = s[:, 0].min()
    print(f"== M = {M}: {len(s)} workgroups; kernel span {(s[:, 16].max() - t0):.0f} ticks; workgroup lifetime "
          f"mean {(s[:, 16] - s[:, 0]).mean():.0f}, max {(s[:, 16] - s[:, 0]).max():.0f}; start offsets: median "
          f"{(sorted(s[:, 0] - t0)[len(s) // 2]):.0f}, max {(s[:, 0] - t0).max():.0f}")
    d = s[:, 1:17] - s[:, 0:16]
    labels = ["x tile -> LDS -> fragments", "loop entry barrier"]
    for c in range(2):
        labels += [f"chunk {c}: commit W1 + barrier", f"chunk {c}: GEMM1 + u -> sG", f"chunk {c}: barrier",
                   f"chunk {c}: pass (gelu; u, g stores) + commit W2", f"chunk {c}: barrier",
                   f"chunk {c}: GEMM2 + loop barrier" if c == 0 else "chunk 1: GEMM2 ... chunks 2-7 ... last GEMM2"]
    labels += ["barrier before the epilogue", "epilogue (bias, dropout, residual, LayerNorm, stores)"]
    for i, lab in enumerate(labels):
        print(f"  {lab:58s} mean {d[:, i].mean():9.0f}  p90 {sorted(d[:, i])[int(0.9 * len(d))]:9.0f} ticks")


for M in (49152, 102400):
    run(M)
