"""The LDS-DMA ring form of the weight-gradient GEMMs (csrc/dw_ring.hip): dW = dy^T x in token slabs + the bias row sums,
against torch on the same bf16 operands -- ragged token counts (a slab's last stage is partial, T below one stage), every
tile count the encoder shapes produce (d_model 128 / 256 / 384), with and without the bias rows, grouped and alone.

The kernel sits behind `xfmr_encoder_bwd` (the parity tests of tests/test_gpu_model.py / test_gpu_packed.py /
test_gpu_fullsize.py run it on every d_model >= 128 shape); this file reaches it through the library's internal entry point
`xf_linear_bwd_dw_group` (extern "C" but not part of include/xfmr_hip.h) to put the slab arithmetic itself under a test."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


class Item(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("N", C.c_int32), ("K", C.c_int32), ("slabs", C.c_void_p),
                ("bias_part", C.c_void_p), ("splits", C.c_void_p)]


@pytest.fixture(scope="module")
def lib():
    from xfmr_rec_amd import _native as N

    lib = N.load()
    lib.xf_linear_bwd_dw_group.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_uint32, C.c_void_p]
    lib.xf_dw_ring_takes.argtypes = [C.c_void_p, C.c_int, C.c_int64, C.c_int32, C.c_uint32]
    lib.xf_dw_ring_takes.restype = C.c_bool
    lib.xf_linear_bwd_dw_slab_bytes.restype = C.c_size_t
    lib.xf_linear_bwd_dw_slab_bytes.argtypes = [C.c_int64, C.c_int32, C.c_int32]
    return lib


def _run(lib, T, shapes, seed=0):
    g = torch.Generator().manual_seed(seed)
    keep, arr = [], (Item * len(shapes))()
    for i, (n, k, bias) in enumerate(shapes):
        dy = (torch.randn(T, n, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        x = (torch.randn(T, k, generator=g) * 0.5).to(torch.bfloat16).to(DEV)
        slabs = torch.full((lib.xf_linear_bwd_dw_slab_bytes(T, n, k) // 4,), float("nan"), device=DEV)
        bpart = torch.full((256 * n,), float("nan"), device=DEV) if bias else None
        splits = (C.c_int * 1)()
        arr[i] = Item(dy.data_ptr(), x.data_ptr(), n, k, slabs.data_ptr(), bpart.data_ptr() if bias else None,
                      C.addressof(splits))
        keep.append((dy, x, n, k, slabs, bpart, splits))
    assert lib.xf_dw_ring_takes(arr, len(shapes), T, 1, 3), "shape not routed to the ring kernel"
    rc = lib.xf_linear_bwd_dw_group(arr, len(shapes), T, 1, 3, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, rc
    torch.cuda.synchronize()
    out = []
    for dy, x, n, k, slabs, bpart, splits in keep:
        s = splits[0]
        dw = slabs[: s * n * k].view(s, n, k).sum(0)
        db = bpart[: s * n].view(s, n).sum(0) if bpart is not None else None
        out.append((dy, x, dw, db, slabs[: s * n * k].clone()))
    return out


@pytest.mark.parametrize("T", [1, 63, 64, 65, 200, 1000, 4097, 25600 + 17])
def test_ring_slabs_equal_torch_on_ragged_token_counts(lib, T):
    shapes = [(128, 512, False), (512, 128, True), (128, 128, False), (384, 128, True)]  # config 2's layer, as grouped
    for dy, x, dw, db, _ in _run(lib, T, shapes):
        ref = dy.float().t() @ x.float()
        assert torch.isfinite(dw).all()
        assert float((dw - ref).norm() / ref.norm().clamp_min(1e-20)) <= 1e-5
        if db is not None:
            rb = dy.float().sum(0)
            assert float((db - rb).norm() / rb.norm().clamp_min(1e-20)) <= 1e-5


@pytest.mark.parametrize("shape", [(768, 256, True), (256, 1024, False), (1024, 256, True), (1152, 384, True), (384, 384, False)])
def test_ring_tile_counts_of_the_wider_models(lib, shape):
    for T in (777, 12800):
        (dy, x, dw, db, raw), = _run(lib, T, [shape])
        ref = dy.float().t() @ x.float()
        assert float((dw - ref).norm() / ref.norm()) <= 1e-5
        if db is not None:
            assert float((db - dy.float().sum(0)).norm() / dy.float().sum(0).norm()) <= 1e-5
        (_, _, _, _, raw2), = _run(lib, T, [shape])
        assert torch.equal(raw, raw2)  # bit-reproducible, slab by slab


def test_shapes_outside_the_ring_keep_the_generic_kernel(lib):
    arr = (Item * 1)()
    t = torch.zeros(64, 192, dtype=torch.bfloat16, device=DEV)
    sp = (C.c_int * 1)()
    arr[0] = Item(t.data_ptr(), t.data_ptr(), 192, 64, t.data_ptr(), None, C.addressof(sp))  # d_model 64: 64-wide weights
    assert not lib.xf_dw_ring_takes(arr, 1, 64, 1, 3)
    arr[0] = Item(t.data_ptr(), t.data_ptr(), 128, 128, t.data_ptr(), None, C.addressof(sp))
    assert not lib.xf_dw_ring_takes(arr, 1, 64, 0, 0)  # fp32 parity policy
    assert lib.xf_dw_ring_takes(arr, 1, 64, 1, 3)
