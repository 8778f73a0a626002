#!/usr/bin/env python3
"""Which segments set the reduction launch's time? xf_multi_rowsum on the encoder backward's real segment set at batch
512 (4 layers: split-K slabs of the four weights, bias partial rows, LayerNorm partial records), whole and in parts."""
import ctypes as C
import pathlib
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT / "transformer-recommenders_amd"))
from xfmr_rec_amd import _native as N  # noqa: E402


class Seg(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld", C.c_int), ("pad", C.c_int)]


DEV = "cuda"
H, I, T = 128, 512, 102400
keep = []


def seg(rows, cols, ld=None, off=0, buf=None):
    ld = ld or cols
    if buf is None:
        buf = torch.randn(rows, ld, device=DEV)
        keep.append(buf)
    dst = torch.empty(cols, device=DEV)
    keep.append(dst)
    return Seg(buf.data_ptr() + 4 * off, dst.data_ptr(), rows, cols, ld, 0)


def layer():
    w = [seg(62, H * I), seg(62, I * H), seg(115, H * H), seg(80, 3 * H * H)]
    b = [seg(62, I), seg(80, 3 * H)]
    ln = []
    for _ in range(2):
        rec = torch.randn(1600, 3 * H, device=DEV)
        keep.append(rec)
        ln += [seg(1600, H, 3 * H, k * H, rec) for k in range(3)]
    return w, b, ln


W, B, LN = [], [], []
for _ in range(4):
    w, b, ln = layer()
    W += w; B += b; LN += ln
fn = N.load().xf_multi_rowsum
fn.restype = C.c_int
fn.argtypes = [C.c_void_p, C.c_int, C.c_void_p]


def timeit(label, segs):
    arr = (Seg * len(segs))(*segs)
    for _ in range(3):
        assert fn(C.cast(arr, C.c_void_p), len(segs), N.stream()) == 0
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        fn(C.cast(arr, C.c_void_p), len(segs), N.stream())
    t1.record()
    torch.cuda.synchronize()
    mb = sum(s.rows * s.cols * 4 for s in segs) / 1e6
    us = t0.elapsed_time(t1) / 20 * 1e3
    print(f"{label:44s} {len(segs):3d} segments {mb:7.1f} MB {us:7.1f} us {mb / us / 1e6:6.2f} TB/s")


timeit("all (as in the step)", W + B + LN)
timeit("weight slabs only", W)
timeit("bias partial rows only", B)
timeit("LayerNorm records only", LN)
timeit("W2 slabs of the 4 layers (62 x 65536)", W[0::4])
timeit("Wo slabs of the 4 layers (115 x 16384)", W[2::4])
