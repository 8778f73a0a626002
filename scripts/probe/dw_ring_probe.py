#!/usr/bin/env python3
"""The weight-gradient GEMMs of one layer in isolation: the LDS-DMA ring kernel (csrc/dw_ring.hip) against the generic
split-K kernel, per weight and as the layer's group, with a value check of the two against each other and against torch.
usage (GPU box): XFMR_DW_RING=0 python scripts/probe/dw_ring_probe.py [T]   (the env var only steers the `generic` legs)"""
import ctypes as C, os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "transformer-recommenders_amd"):
    sys.path.insert(0, str(p))
os.environ["XFMR_DW_RING"] = "0"  # xf_linear_bwd_dw_group = the generic kernel here; the ring is called directly
import torch
from xfmr_rec_amd import _native as N

lib = N.load()
class Item(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32), ("slabs", C.c_void_p),
                ("bias_part", C.c_void_p), ("splits", C.c_void_p)]
lib.xf_linear_bwd_dw_group.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_uint32, C.c_void_p]
lib.xf_dw_ring_launch.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_void_p]
lib.xf_linear_bwd_dw_slab_bytes.restype = C.c_size_t
lib.xf_linear_bwd_dw_slab_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]

T = int(sys.argv[1]) if len(sys.argv) > 1 else 102400
H, I = 128, 512
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
mk = lambda n: (torch.randn(T, n, device=dev, generator=g) * 0.5).to(torch.bfloat16)
shapes = {"qkv": (3 * H, H, True), "out": (H, H, False), "ffn1": (I, H, True), "ffn2": (H, I, False)}
ops = {}
for name, (n, k, bias) in shapes.items():
    dy, x = mk(n), mk(k)
    slab_bytes = lib.xf_linear_bwd_dw_slab_bytes(T, n, k)
    slabs = [torch.zeros(slab_bytes // 4, device=dev) for _ in range(2)]
    bparts = [torch.zeros(256 * n, device=dev) if bias else None for _ in range(2)]
    splits = [(C.c_int * 1)(), (C.c_int * 1)()]  # (the two kernels have their own slab plans)
    ops[name] = (dy, x, n, k, slabs, bparts, splits)

def items(names, which):
    arr = (Item * len(names))()
    for i, nm in enumerate(names):
        dy, x, n, k, slabs, bparts, splits = ops[nm]
        arr[i] = Item(dy.data_ptr(), x.data_ptr(), n, k, slabs[which].data_ptr(),
                      bparts[which].data_ptr() if bparts[which] is not None else None, C.addressof(splits[which]))
    return arr

def run(names, ring):
    arr = items(names, 1 if ring else 0)
    st = torch.cuda.current_stream().cuda_stream
    rc = lib.xf_dw_ring_launch(arr, len(names), T, st) if ring else lib.xf_linear_bwd_dw_group(arr, len(names), T, 1, 3, st)
    assert rc == 0, rc

def timeit(fn, reps=20):
    for _ in range(3): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

for names in (["qkv"], ["out"], ["ffn1"], ["ffn2"], ["ffn2", "ffn1", "out", "qkv"]):
    byts = sum(T * (ops[n][2] + ops[n][3]) * 2 for n in names)
    tg, tr = timeit(lambda: run(names, False)), timeit(lambda: run(names, True))
    print(f"{'+'.join(names):22s} generic {tg:7.1f} us {byts / tg / 1e6:5.2f} TB/s   ring {tr:7.1f} us {byts / tr / 1e6:5.2f} TB/s")
torch.cuda.synchronize()
for nm, (dy, x, n, k, slabs, bparts, splits) in ops.items():
    sg, s = splits[0][0], splits[1][0]
    a = slabs[0][: sg * n * k].view(sg, n, k).sum(0)
    b = slabs[1][: s * n * k].view(s, n, k).sum(0)
    ref = dy.float().t() @ x.float()
    rel = lambda u, v: float((u - v).norm() / v.norm())
    msg = f"{nm}: slabs {sg} / {s} generic-vs-torch {rel(a, ref):.2e} ring-vs-torch {rel(b, ref):.2e}"
    if bparts[0] is not None:
        ba, bb = bparts[0][: sg * n].view(sg, n).sum(0), bparts[1][: s * n].view(s, n).sum(0)
        msg += f"  bias: generic {rel(ba, dy.float().sum(0)):.2e} ring {rel(bb, dy.float().sum(0)):.2e}"
    print(msg)
