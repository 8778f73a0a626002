#!/usr/bin/env python3
"""Where does ffn_fwd_fused_kernel's time go? Times xf_ffn_fwd_fused_ex at the benchmark's row count with pieces of its
work switched off through its own arguments (null outputs skip the u / g stores; dropout off skips the mask hash).
    python scripts/probe/ffn_fwd_probe.py            (on the GPU box)"""
import ctypes as C
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "transformer-recommenders_amd"))
from xfmr_rec_amd import _native as N  # noqa: E402

lib = N.load()
DEV = "cuda"
M, H, I = 102400, 128, 512
g = torch.Generator().manual_seed(0)
x16 = torch.randn(M, H, generator=g).to(DEV).to(torch.bfloat16)
w1 = (torch.randn(I, H, generator=g) * 0.08).to(DEV).to(torch.bfloat16)
b1 = (0.1 * torch.randn(I, generator=g)).to(DEV)
w2 = (torch.randn(H, I, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
b2 = (0.1 * torch.randn(H, generator=g)).to(DEV)
res = torch.randn(M, H, generator=g).to(DEV)
gamma = torch.ones(H, device=DEV)
beta = torch.zeros(H, device=DEV)
u = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
gg = torch.empty(M, I, device=DEV, dtype=torch.bfloat16)
pre = torch.empty(M, H, device=DEV)
y = torch.empty(M, H, device=DEV)
y16 = torch.empty(M, H, device=DEV, dtype=torch.bfloat16)
mean = torch.empty(M, device=DEV)
rstd = torch.empty(M, device=DEV)
fn = lib.xf_ffn_fwd_fused_ex
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_float, N.Seed, C.c_uint32,
                                  C.c_void_p, C.c_void_p, C.c_float] + [C.c_void_p] * 5


def run(store_u=True, store_g=True, p=0.1, y16_out=True, rows=M):
    return fn(N.ptr(x16), N.ptr(w1), N.ptr(b1), N.ptr(w2), N.ptr(b2), N.ptr(u) if store_u else None,
              N.ptr(gg) if store_g else None, N.ptr(pre), rows, H, I, N.ptr(res), p, 5, 9, N.ptr(gamma), N.ptr(beta), 1e-12,
              N.ptr(y), N.ptr(y16) if y16_out else None, N.ptr(mean), N.ptr(rstd), N.stream())


def timeit(label, **kw):
    for _ in range(3):
        assert run(**kw) == 0
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        run(**kw)
    t1.record()
    torch.cuda.synchronize()
    print(f"{label:48s} {t0.elapsed_time(t1) / 20 * 1e3:8.1f} us")


timeit("full (u, g stored; dropout 0.1)")
timeit("no dropout", p=0.0)
timeit("u not stored", store_u=False)
timeit("u, g not stored", store_u=False, store_g=False)
timeit("u, g, y16 not stored; no dropout", store_u=False, store_g=False, y16_out=False, p=0.0)
for rows in (98304, 49152, 24576):  # 1536 / 768 / 384 tiles: 2.0 / 1.0 / 0.5 per workgroup slot at 3 per CU
    timeit(f"full, {rows} rows ({rows // 64} tiles)", rows=rows)
