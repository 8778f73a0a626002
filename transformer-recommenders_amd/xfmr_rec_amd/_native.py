"""ctypes binding of ``libxfmr_hip.so`` (C ABI declared in ``include/xfmr_hip.h``).

There is no CPU fallback: every op in this package goes through this library. If the shared object is
missing, cannot be loaded, or a tensor is not on a HIP device, the call raises -- it never silently
computes with PyTorch instead.
"""

from __future__ import annotations

import ctypes as C
import os
import pathlib

import torch

_HERE = pathlib.Path(__file__).resolve().parent
# XFMR_HIP_LIB points at another build of the same library (kernel experiments); the default is the in-tree one
LIB_PATH = pathlib.Path(os.environ["XFMR_HIP_LIB"]) if os.environ.get("XFMR_HIP_LIB") else _HERE / "libxfmr_hip.so"

PREC_F32, PREC_BF16 = 0, 1
PRECISIONS = {"fp32": PREC_F32, "f32": PREC_F32, "bf16": PREC_BF16}
EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_DROP_RES = 0, 1, 2
NEG_SHARED, NEG_CATALOG = 0, 1
ATTN_CAUSAL, ATTN_BIDIRECTIONAL = 0, 1  # xfmr_attn_{fwd,bwd}_mode
# xfmr_encoder_cfg.flags
ENC_BIDIRECTIONAL, ENC_LN_UNFUSED, ENC_FFN_UNFUSED, ENC_FFN_BWD_UNFUSED, ENC_DW_INLINE, ENC_DW_SIDE_ANY = 1, 2, 4, 8, 16, 32
ENC_DW_UNPAIRED, ENC_REDUCE_HALF_EARLY = 64, 128
LOSS_DTOK_ZEROED = 1  # xfmr_loss_cfg.flags
ABI_VERSION = 3
ECOMM = -6  # XFMR_ECOMM: RCCL not loadable / an RCCL call failed (xfmr_comm_last_error has RCCL's text)
NUM_LOSSES, NUM_STATS = 7, 16
LOSS_IDS = {
    "AlignmentLoss": 0,
    "AlignmentContrastiveLoss": 1,
    "ContrastiveLoss": 2,
    "InfoNCELoss": 3,
    "NCELoss": 4,
    "PairwiseHingeLoss": 5,
    "PairwiseLogisticLoss": 6,
}
STAT = dict(
    n_valid=0, n_query=1, neg_density=2, pos_mean=3, pos_std=4, pos_min=5, pos_max=6,
    neg_mean=7, neg_std=8, neg_min=9, neg_max=10, neg_count=11, neg_distinct=12, pos_density=13, attn_density=14,
)


class EncoderCfg(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("seq_len", C.c_int32), ("hidden", C.c_int32), ("heads", C.c_int32),
        ("inter", C.c_int32), ("layers", C.c_int32), ("max_pos", C.c_int32), ("precision", C.c_int32),
        ("ln_eps", C.c_float), ("hidden_dropout", C.c_float), ("attn_dropout", C.c_float),
        ("flags", C.c_uint32), ("seed", C.c_uint64),
        ("step_device", C.c_void_p), ("embed_event", C.c_void_p), ("context", C.c_void_p),
        ("grads_half_event", C.c_void_p),
        ("profile_kernel", C.c_int32), ("profile_layer", C.c_int32), ("profile_events", C.c_void_p * 2),
        # ABI 3: packed rows (all NULL / 0 = the padded (B, L) layout)
        ("seq_offsets", C.c_void_p), ("row_pos", C.c_void_p), ("packed_rows", C.c_int64),
    ]


PROF_FFN_FWD, PROF_FFN_BWD, PROF_ATTN_FWD, PROF_ATTN_BWD, PROF_DW, PROF_REDUCE = 1, 2, 3, 4, 5, 6


class LossCfg(C.Structure):
    _fields_ = [
        ("train_head", C.c_int32), ("all_heads", C.c_int32), ("mask_false_negatives", C.c_int32),
        ("mode", C.c_int32), ("precision", C.c_int32), ("scale", C.c_float), ("margin", C.c_float),
        ("num_hard_negatives", C.c_int32), ("flags", C.c_uint32),
        ("profile_grad", C.c_void_p * 2), ("profile_log", C.c_void_p * 2),
        ("padded_positions", C.c_int64),  # ABI 3
    ]


class Seed(C.Structure):
    """csrc/common.h XfSeed, by value: the dropout seed argument of the library's INTERNAL entry points (xf_*_ex; tests and
    probes call a few of them directly). A plain int converts to (seed, no device-side counter)."""

    _fields_ = [("seed", C.c_uint64), ("dyn", C.c_void_p)]

    @classmethod
    def from_param(cls, v):
        return v if isinstance(v, cls) else cls(int(v), None)


TARGET_FIRST, TARGET_DIAGONAL, TARGET_EXPLICIT = 0, 1, 2
POOL_MODES = {"mean": 0, "max": 1, "cls": 2, "lasttoken": 3}

_P = C.c_void_p
_SIGNATURES = {
    "xfmr_strerror": (C.c_char_p, [C.c_int]),
    "xfmr_abi_version": (C.c_int, []),
    "xfmr_low_priority_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "xfmr_context_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "xfmr_context_destroy": (C.c_int, [_P]),
    "xfmr_stream_destroy": (C.c_int, [_P]),
    "xfmr_stream_create": (C.c_int, [C.POINTER(C.c_void_p)]),
    "xfmr_event_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int32]),
    "xfmr_event_destroy": (C.c_int, [_P]),
    "xfmr_event_record": (C.c_int, [_P, _P]),
    "xfmr_stream_wait_event": (C.c_int, [_P, _P]),
    "xfmr_event_elapsed_ms": (C.c_int, [_P, _P, C.POINTER(C.c_float)]),
    "xfmr_event_synchronize": (C.c_int, [_P]),
    "xfmr_event_query": (C.c_int, [_P]),
    "xfmr_batch_upload": (C.c_int, [_P, _P, C.c_size_t, _P, _P, _P]),
    "xfmr_param_count": (C.c_int64, [C.POINTER(EncoderCfg)]),
    "xfmr_param_half_offset": (C.c_int64, [C.POINTER(EncoderCfg)]),
    "xfmr_param_offsets": (C.c_int32, [C.POINTER(EncoderCfg), C.POINTER(C.c_int64), C.c_int32]),
    "xfmr_embed_ln_fwd": (C.c_int, [_P, _P, C.c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_float, C.c_float, C.c_uint64, C.c_uint32, _P]),
    "xfmr_pack_rows": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "xfmr_pack_rows_ordered": (C.c_int, [_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int64, _P, _P, _P, _P, _P, _P]),
    "xfmr_embed_param_grads": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "xfmr_layernorm_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "xfmr_layernorm_bwd_workspace": (C.c_size_t, [C.c_int64, C.c_int32]),
    "xfmr_layernorm_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float,
                                     C.c_uint64, C.c_uint32, _P, _P]),
    "xfmr_linear_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_float,
                                  C.c_uint64, C.c_uint32, C.c_int32, _P]),
    "xfmr_linear_bwd_dx": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, _P, _P, C.c_int32, _P]),
    "xfmr_linear_bwd_dw_workspace": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "xfmr_linear_bwd_dw": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, _P, C.c_size_t, _P]),
    "xfmr_colsum_workspace": (C.c_size_t, [C.c_int64, C.c_int32]),
    "xfmr_colsum": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P]),
    "xfmr_attn_fwd": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint64,
                                C.c_uint32, C.c_int32, _P]),
    "xfmr_attn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                C.c_uint64, C.c_uint32, C.c_int32, _P]),
    "xfmr_attn_fwd_mode": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_uint64,
                                     C.c_uint32, C.c_int32, C.c_int32, _P]),
    "xfmr_attn_bwd_mode": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                     C.c_uint64, C.c_uint32, C.c_int32, C.c_int32, _P]),
    "xfmr_encoder_workspace_bytes": (C.c_size_t, [C.POINTER(EncoderCfg)]),
    "xfmr_encoder_fwd": (C.c_int, [C.POINTER(EncoderCfg), _P, _P, _P, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    "xfmr_encoder_bwd": (C.c_int, [C.POINTER(EncoderCfg), _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "xfmr_mean_pool": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P]),
    "xfmr_pool": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "xfmr_l2_normalize_fwd": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "xfmr_l2_normalize_bwd": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_int32, C.c_float, _P]),
    "xfmr_sampled_loss_workspace": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int64]),
    "xfmr_sampled_loss_workspace_cfg": (C.c_size_t, [C.POINTER(LossCfg), C.c_int64, C.c_int32, C.c_int64]),
    "xfmr_sampled_loss": (C.c_int, [C.POINTER(LossCfg), _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32,
                                    _P, _P, _P, _P, C.c_size_t, _P]),
    "xfmr_sampled_loss_prepare": (C.c_int, [C.POINTER(LossCfg), _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_int32, _P,
                                            C.c_size_t, _P]),
    "xfmr_sampled_loss_prepared": (C.c_int, [C.POINTER(LossCfg), _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64,
                                             C.c_int32, _P, _P, _P, _P, C.c_size_t, _P]),
    "xfmr_sampled_loss_lists_workspace": (C.c_size_t, [C.c_int64, C.c_int64, C.c_int32, C.c_int64]),
    "xfmr_sampled_loss_lists_workspace_cfg": (C.c_size_t, [C.POINTER(LossCfg), C.c_int64, C.c_int64, C.c_int32,
                                                           C.c_int64]),
    "xfmr_sampled_loss_lists": (C.c_int, [C.POINTER(LossCfg), _P, _P, _P, C.c_int64, C.c_int64, _P, _P, _P, C.c_int64,
                                          C.c_int32, _P, _P, _P, _P, C.c_size_t, _P]),
    "xfmr_dense_loss_workspace": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "xfmr_dense_loss": (C.c_int, [C.POINTER(LossCfg), _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P, _P,
                                  _P, _P, C.c_size_t, _P]),
    "xfmr_dense_loss_grads": (C.c_int, [C.POINTER(LossCfg), _P, _P, _P, C.c_int32, C.c_int64, C.c_int32, C.c_int32, _P,
                                        _P, _P, _P, _P, C.c_size_t, _P]),
    "xfmr_table_rnorm": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "xfmr_table_prepare": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int32, _P]),
    "xfmr_seq_sample_workspace": (C.c_size_t, [C.c_int32, C.c_int64]),
    "xfmr_seq_sample": (C.c_int, [_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_int32,
                                  C.c_uint64, _P, _P, _P, _P, C.c_size_t, _P]),
    "xfmr_topk_workspace": (C.c_size_t, [C.c_int64, C.c_int64]),
    "xfmr_topk": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int64, C.c_int32, _P, _P, C.c_int32, C.c_int32, _P, _P, _P,
                            C.c_size_t, _P]),
    "xfmr_retrieval_metrics": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "xfmr_adamw": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                             C.c_int64, C.c_float, _P]),
    "xfmr_adamw_dev": (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                 _P, C.c_int32, C.c_float, _P]),
    "xfmr_step_advance": (C.c_int, [_P, _P]),
    "xfmr_scale_by_device_scalar": (C.c_int, [_P, C.c_int64, _P, _P]),
    "xfmr_selftest_mfma": (C.c_int, [_P, _P]),
    "xfmr_comm_unique_id": (C.c_int, [_P]),
    "xfmr_comm_create": (C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    "xfmr_comm_destroy": (C.c_int, [_P]),
    "xfmr_allreduce_flat": (C.c_int, [_P, _P, C.c_int64, _P]),
    "xfmr_comm_last_error": (C.c_char_p, []),
}
COMM_ID_BYTES = 128
EXPORTED_SYMBOLS = tuple(_SIGNATURES)

_lib = None


class NativeLibraryError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load the library once. ``torch`` is imported first so that its bundled HIP runtime
    (same soname ``libamdhip64.so.7``) is the one the library binds to."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise NativeLibraryError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C transformer-recommenders_amd/csrc`). There is no CPU fallback."
        )
    try:
        lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    except OSError as e:  # pragma: no cover - depends on the host
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    got = lib.xfmr_abi_version() if hasattr(lib, "xfmr_abi_version") else None
    if got != ABI_VERSION:  # checked before binding the symbols: an older build fails here with a version message
        raise NativeLibraryError(f"{LIB_PATH}: ABI version {got}, this package binds version {ABI_VERSION}: rebuild the "
                                 "library (python -c 'import __graft_entry__ as g; g.build()')")
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().xfmr_strerror(rc).decode()
        if rc == ECOMM:
            msg += ": " + load().xfmr_comm_last_error().decode()
        raise RuntimeError(f"{what} failed: {msg} (code {rc})")


def ptr(t: torch.Tensor | None) -> int | None:
    """Device pointer of a contiguous HIP tensor (None passes NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(
            "xfmr_rec_amd ops run only on a HIP device (MI355X); got a CPU tensor. There is no CPU fallback."
        )
    if not t.is_contiguous():
        raise RuntimeError("xfmr_rec_amd ops need contiguous tensors")
    return t.data_ptr()


def stream() -> int:
    """hipStream_t of torch's current stream on the current device. Called once per launch sequence: the raw accessor
    (no ``torch.cuda.Stream`` object, no ``is_available()`` / device-count probe per call: ~10 us each on the host of a
    host-bound small step) where this torch has it."""
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:  # pragma: no cover - older / newer torch without the private accessors
        return torch.cuda.current_stream().cuda_stream


def precision_id(p) -> int:
    if isinstance(p, int):
        return p
    try:
        return PRECISIONS[str(p).lower()]
    except KeyError:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}; got {p!r}") from None
