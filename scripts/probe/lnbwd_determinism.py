#!/usr/bin/env python3
"""Diagnostic (not a test): run the LayerNorm-fused dX GEMM (xf_linear_bwd_dx_lnbwd_ex) and the LayerNorm-fused forward
GEMM (xf_linear_ln_fwd_ex) several times on identical inputs and report WHICH outputs differ between launches and where
(row inside the 64-row tile, column, wave / half strip / pass of the epilogue's lane map). Used with
XFMR_HIP_LIB=<a build with the default launch bound> to localise the run-to-run differences DESIGN.md section 4 records."""
import ctypes as C
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "transformer-recommenders_amd"):
    sys.path.insert(0, str(p))
import torch  # noqa: E402

from xfmr_rec_amd import _native as N  # noqa: E402

DEV = "cuda"
lib = N.load()
print("library:", N.LIB_PATH)
M, Nn, K = 102400, 512, 128  # dy (M, Nn) x w (Nn, K) -> dx (M, 128)
g = torch.Generator().manual_seed(0)
dy = torch.randn(M, Nn, generator=g).to(DEV).to(torch.bfloat16)
w = (torch.randn(Nn, K, generator=g) * 0.05).to(DEV).to(torch.bfloat16)
rg = torch.randn(M, K, generator=g).to(DEV)
lnx = torch.randn(M, K, generator=g).to(DEV)
mean = lnx.mean(-1).contiguous()
rstd = (lnx.var(-1, unbiased=False) + 1e-12).rsqrt().contiguous()
gamma = (1 + 0.1 * torch.randn(K, generator=g)).to(DEV)

fn = lib.xf_linear_bwd_dx_lnbwd_ex
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32] + [C.c_void_p] * 5 + [
    C.c_float, N.Seed, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int32, C.c_uint32,
    C.c_void_p, C.c_float, C.c_uint32]


PROD = C.CDLL(str(ROOT / "transformer-recommenders_amd" / "xfmr_rec_amd" / "libxfmr_hip.so"))  # the in-tree build
fn_prod = PROD.xf_linear_bwd_dx_lnbwd_ex
fn_prod.restype, fn_prod.argtypes = fn.restype, fn.argtypes


DBG = None
if hasattr(lib, "xf_diag_set_buffer"):  # a build with XF_LN_DIAG & 4: per (row, column half) intermediate values
    DBG = torch.zeros(M, 2, 8, device=DEV)
    lib.xf_diag_set_buffer.restype = None
    lib.xf_diag_set_buffer.argtypes = [C.c_void_p]
    lib.xf_diag_set_buffer(N.ptr(DBG))


lib.xf_ln_row_tiles.restype, lib.xf_ln_row_tiles.argtypes = C.c_int, [C.c_int64]
N_TILES = max(lib.xf_ln_row_tiles(M), PROD.xf_ln_row_tiles(M)) if hasattr(lib, "xf_ln_row_tiles") else (M + 63) // 64


def run_bwd(p_drop, fn=fn):
    dx = torch.empty(M, K, device=DEV)
    d16 = torch.empty(M, K, device=DEV, dtype=torch.bfloat16)
    parts = torch.zeros(N_TILES, 3, K, device=DEV)
    blocks = C.c_int(0)
    rc = fn(N.ptr(dy), N.ptr(w), M, Nn, K, N.ptr(rg), N.ptr(lnx), N.ptr(mean), N.ptr(rstd), N.ptr(gamma), p_drop, 5, 9,
            N.ptr(dx), N.ptr(d16), N.ptr(parts), C.byref(blocks), N.precision_id("bf16"), 3, N.stream(), 0.0, 0)
    assert rc == 0, rc
    torch.cuda.synchronize()
    return dx, d16.float(), parts


def report(name, a, b):
    d = (a != b)
    n = int(d.sum())
    print(f"  {name}: {n} elements differ of {a.numel()}")
    if n and a.dim() == 2 and a.shape[0] == M:
        rows = d.any(-1).nonzero().flatten()
        print(f"    rows affected: {rows.numel()}; first: {rows[:12].tolist()}")
        r64 = rows % 64
        print(f"    row % 64 histogram: {torch.bincount(r64, minlength=64).tolist()}")
        cols = d[rows].float().sum(0)
        print(f"    differing columns per affected row: min {int(d[rows].sum(-1).min())} max {int(d[rows].sum(-1).max())}; "
              f"column histogram (by 16): {cols.view(8, 16).sum(-1).tolist()}")
        r = int(rows[0])
        rel = ((a[r] - b[r]).abs() / b[r].abs().clamp(min=1e-6)).max().item()
        print(f"    row {r}: max rel diff {rel:.3e}")
    elif n:
        idx = d.nonzero()[:8].tolist()
        print(f"    first indices: {idx}")


def explain(dx_bad, dx_good, rows):
    """d = rstd * (gg - mg - h * mgx) with mg = (sum gg) / 128, mgx = (sum gg h) / 128, the two sums exchanged between the
    two waves that hold a row's column halves. From a differing row: least squares for (delta mg, delta mgx), printed beside
    each half's share of the sums -- a missing / stale partner record shows up as exactly one half's share."""
    for r in rows[:6].tolist():
        dyr = dy[r].float() @ w.float() + rg[r]           # row of dY W + residual gradient (fp32)
        h = (lnx[r] - mean[r]) * rstd[r]
        gg = dyr * gamma
        delta = (dx_bad[r] - dx_good[r]) / rstd[r]        # = -(d_mg + h d_mgx)
        A = torch.stack([torch.ones_like(h), h], 1)
        sol = torch.linalg.lstsq(A.double().cpu(), (-delta).double().cpu()[:, None]).solution.flatten()
        res = float(((A.double().cpu() @ sol[:, None]).flatten() + delta.double().cpu()).abs().max())
        s1 = [float(gg[:64].sum()) / 128, float(gg[64:].sum()) / 128]
        s2 = [float((gg * h)[:64].sum()) / 128, float((gg * h)[64:].sum()) / 128]
        if DBG is not None and float(DBG[r, 0, 4]) == 0.0 and float(DBG[r, 0, 5]) == 0.0:  # light dump (XF_LN_DIAG & 16)
            ggh = gg * h
            for wc in (0, 1):
                d = DBG[r, wc].tolist()
                print(f"      wave wc={wc}: used sum(gg) = {d[0]:+.5f} [ref {float(gg.sum()):+.5f}], sum(gg h) = {d[1]:+.5f} "
                      f"[ref {float(ggh.sum()):+.5f}], rs {d[2]:+.6f} [ref {float(rstd[r]):+.6f}], h[{64 * wc}] {d[3]:+.6f} "
                      f"[ref {float(h[64 * wc]):+.6f}]")
        elif DBG is not None:
            ggh = gg * h
            want = [[float(gg[:64].sum()), float(ggh[:64].sum())], [float(gg[64:].sum()), float(ggh[64:].sum())]]
            for wc in (0, 1):
                d = DBG[r, wc].tolist()
                print(f"      wave wc={wc}: own (s1, s2) = ({d[0]:+.5f}, {d[1]:+.5f}) [fp32 reference ({want[wc][0]:+.5f}, "
                      f"{want[wc][1]:+.5f})]; read back r0 = ({d[2]:+.5f}, {d[3]:+.5f}) r1 = ({d[4]:+.5f}, {d[5]:+.5f}); "
                      f"mu {d[6]:+.6f} (ref {float(mean[r]):+.6f}) rs {d[7]:+.6f} (ref {float(rstd[r]):+.6f})")
        print(f"    row {r} (row%64={r % 64}): d_mg={sol[0]:+.5f} d_mgx={sol[1]:+.5f} (fit residual {res:.1e}); "
              f"shares of mg by column half: {s1[0]:+.5f} {s1[1]:+.5f}; of mgx: {s2[0]:+.5f} {s2[1]:+.5f}")


for p_drop in (0.0, 0.1):
    print(f"== dX + LayerNorm backward epilogue, dropout {p_drop}")
    ref = run_bwd(p_drop, fn_prod)  # the production build (deterministic: checked below) is the reference
    assert all(torch.equal(x, y) for x, y in zip(ref, run_bwd(p_drop, fn_prod)))
    if "--self" in sys.argv:  # builds whose rounding differs from the production build: majority vote of 5 own launches
        runs = [run_bwd(p_drop) for _ in range(5)]
        ref = tuple(torch.stack([r_[k] for r_ in runs]).median(0).values for k in range(3))
    bad = 0
    for it in range(16):
        out = run_bwd(p_drop)
        if not all(torch.equal(x, y) for x, y in zip(ref, out)):
            bad += 1
            if bad <= 2:
                print(f" launch {it}:")
                for name, x, y in zip(("dx", "d_lin16", "partials"), out, ref):
                    report(name, x, y)
                if p_drop == 0.0:
                    explain(out[0], ref[0], (out[0] != ref[0]).any(-1).nonzero().flatten())
    print(f" {bad} of 16 launches differ from the first")
