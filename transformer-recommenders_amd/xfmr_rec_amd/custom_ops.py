"""``torch.ops.xfmr.*``: the hot path's kernels registered as PyTorch custom operators (``torch.library.custom_op``).

The reference's operator API for this path is ``nn.Module`` + autograd (``xfmr_rec/models.py:306-345``,
``xfmr_rec/losses.py:128-155``). This module gives the C ABI (``include/xfmr_hip.h``) that shape for PyTorch's own
machinery: every op has a schema, a fake (shape-only) implementation, and -- where the reference differentiates -- an
autograd formula whose backward is again a registered op, so ``torch.compile`` / ``torch.export`` of a model that
contains them sees opaque, shape-inferable calls instead of ``ctypes`` and neither graph-breaks nor fails:

    xfmr::encoder            gather + BertEmbeddings + N x BertLayer  (models.py:336-345, TF:modeling_bert.py:68-448)
    xfmr::encoder_bwd        its backward (the flat gradient of every trainable tensor)
    xfmr::sampled_loss       compute_embeds + the seven heads + LogitsStatistics on (B*L) positions (trainer.py:250-264)
    xfmr::sampled_loss_lists EmbedLoss.forward on [positive | shared negatives] / full-catalogue candidates
    xfmr::dense_loss         EmbedLoss.forward on a dense (N,C,H) candidate tensor (losses.py:128-155)
    xfmr::scale_by_device_scalar_   d *= g (the upstream gradient of a loss, a device scalar)
    xfmr::l2_normalize(_bwd) F.normalize(x, dim=-1) (models.py:393-394)
    xfmr::pool               sentence-transformers Pooling (models.py:143-147)
    xfmr::adamw_             torch.optim.AdamW step over the flat buffer (trainer.py:327-332)

The arithmetic is the library's; the registration adds no computation. Host handles (HIP events, the xfmr_context, a
device step counter) travel as plain integers / an optional tensor, mirroring ``xfmr_encoder_cfg`` / ``xfmr_loss_cfg``.

Cost of the dispatcher: a Python-registered op with an autograd formula takes ~130 us of host time per call where a
``torch.autograd.Function`` takes ~12 (measured, torch 2.10) -- 8 calls per step would make every batch below ~256 x 200
host-bound. The EAGER training step therefore enters the same implementations through the ``autograd.Function`` s of
:mod:`ops` (``ops.encoder`` / ``ops.sampled_loss_train`` pick by ``torch.compiler.is_compiling()``; ``XFMR_TORCH_OPS=1``
forces the registered ops in eager mode too -- the two routes are compared bit for bit in tests/test_gpu_custom_ops.py).
"""

from __future__ import annotations

import ctypes as C
from typing import List, Optional, Tuple

import torch
from torch import Tensor

from . import _native as N

_PREC_NAME = {v: k for k, v in N.PRECISIONS.items()}
_U64 = (1 << 64) - 1
N_HANDLES = 7  # embed_event, context, grads_half_event, profile_kernel, profile_layer, profile_event0, profile_event1


def _signed64(x: int) -> int:
    x &= _U64
    return x - (1 << 64) if x >= (1 << 63) else x


def _encoder_cfg(B, L, H, heads, inter, layers, max_pos, precision, ln_eps, hidden_dropout, attn_dropout, flags, seed,
                 step_device, handles, seq_offsets=None, row_pos=None) -> N.EncoderCfg:
    h = [int(x) & _U64 for x in handles] + [0] * (N_HANDLES - len(handles))
    return N.EncoderCfg(
        batch=int(B), seq_len=int(L), hidden=int(H), heads=heads, inter=inter, layers=layers, max_pos=max_pos,
        precision=N.precision_id(precision), ln_eps=ln_eps, hidden_dropout=hidden_dropout, attn_dropout=attn_dropout,
        flags=int(flags) & 0xFFFFFFFF, seed=int(seed) & _U64,
        step_device=step_device.data_ptr() if step_device is not None else None,
        embed_event=h[0] or None, context=h[1] or None, grads_half_event=h[2] or None,
        profile_kernel=h[3], profile_layer=h[4], profile_events=(C.c_void_p * 2)(h[5] or None, h[6] or None),
        seq_offsets=seq_offsets.data_ptr() if seq_offsets is not None else None,
        row_pos=row_pos.data_ptr() if row_pos is not None else None,
        packed_rows=int(row_pos.numel()) if row_pos is not None else 0,
    )


def _batch_dims(item_idx, seq_offsets, packed_seq_len):
    """(B, L) of the call: the index tensor's shape, or -- packed rows -- the offset table's length and the given seq_len."""
    if seq_offsets is None:
        B, L = item_idx.shape
        return B, L
    return seq_offsets.shape[0] - 1, packed_seq_len


# ------------------------------------------------------------------------------------------------ encoder
@torch.library.custom_op("xfmr::encoder", mutates_args=())
def encoder(flat_params: Tensor, item_idx: Tensor, table: Tensor, heads: int, inter: int, layers: int, max_pos: int,
            precision: str, ln_eps: float, hidden_dropout: float, attn_dropout: float, flags: int, seed: int,
            step_device: Optional[Tensor], handles: List[int], seq_offsets: Optional[Tensor], row_pos: Optional[Tensor],
            packed_seq_len: int) -> Tuple[Tensor, Tensor, Tensor]:
    """(token_embeddings (B,L,H) f32, key_mask (B,L) u8, saved activations (bytes) u8). Packed rows (seq_offsets (B+1) i32,
    row_pos (rows) i32, packed_seq_len = L; ops.pack_rows): item_idx (rows,) -> token_embeddings (rows,H), key_mask (rows)."""
    from . import ops

    B, L = _batch_dims(item_idx, seq_offsets, packed_seq_len)
    cfg = _encoder_cfg(B, L, table.shape[1], heads, inter, layers, max_pos, precision, ln_eps, hidden_dropout,
                       attn_dropout, flags, seed, step_device, handles, seq_offsets, row_pos)
    return ops.encoder_fwd(cfg, flat_params, item_idx, table)


@encoder.register_fake
def _(flat_params, item_idx, table, heads, inter, layers, max_pos, precision, ln_eps, hidden_dropout, attn_dropout, flags,
      seed, step_device, handles, seq_offsets, row_pos, packed_seq_len):
    B, L = _batch_dims(item_idx, seq_offsets, packed_seq_len)
    H = table.shape[1]
    cfg = _encoder_cfg(B, L, H, heads, inter, layers, max_pos, precision, ln_eps, hidden_dropout, attn_dropout, flags, seed,
                       None, handles)
    nbytes = N.load().xfmr_encoder_workspace_bytes(C.byref(cfg))  # a host-side size computation: no device needed
    torch._check(nbytes > 0, lambda: "xfmr::encoder: unsupported encoder configuration")
    rows = (row_pos.shape[0],) if seq_offsets is not None else (B, L)
    return (flat_params.new_empty((*rows, H), dtype=torch.float32), flat_params.new_empty(rows, dtype=torch.uint8),
            flat_params.new_empty((max(int(nbytes), 16),), dtype=torch.uint8))


@torch.library.custom_op("xfmr::encoder_bwd", mutates_args=("d_tok",))
def encoder_bwd(flat_params: Tensor, d_tok: Tensor, key_mask: Tensor, acts: Tensor, heads: int, inter: int, layers: int,
                max_pos: int, precision: str, ln_eps: float, hidden_dropout: float, attn_dropout: float, flags: int,
                seed: int, step_device: Optional[Tensor], handles: List[int], seq_offsets: Optional[Tensor],
                row_pos: Optional[Tensor], packed_seq_len: int) -> Tensor:
    """Flat gradient of the encoder's trainable tensors; ``d_tok`` (B,L,H) -- (rows,H) packed -- is used as scratch."""
    from . import ops

    H = d_tok.shape[-1]
    B, L = (d_tok.shape[0], d_tok.shape[1]) if seq_offsets is None else (seq_offsets.shape[0] - 1, packed_seq_len)
    cfg = _encoder_cfg(B, L, H, heads, inter, layers, max_pos, precision, ln_eps, hidden_dropout, attn_dropout, flags, seed,
                       step_device, handles, seq_offsets, row_pos)
    return ops.encoder_bwd(cfg, flat_params, d_tok, key_mask, acts)


@encoder_bwd.register_fake
def _(flat_params, d_tok, key_mask, acts, *a):
    return torch.empty_like(flat_params)


def _encoder_setup(ctx, inputs, output):
    flat_params, _idx, _table, *scalars = inputs
    _tok, key_mask, acts = output
    step_device, handles, seq_offsets, row_pos, packed_seq_len = scalars[-5:]
    ctx.scalars = scalars[:-5] + [handles]
    ctx.packed_seq_len = packed_seq_len
    ctx.have = (step_device is not None, seq_offsets is not None)
    ctx.save_for_backward(flat_params, key_mask, acts, *[t for t in (step_device, seq_offsets, row_pos) if t is not None])
    ctx.set_materialize_grads(False)


def _encoder_backward(ctx, d_tok, _d_mask, _d_acts):
    n_in = 18
    if d_tok is None:
        return (None,) * n_in
    flat_params, key_mask, acts, *rest = ctx.saved_tensors
    step_device = rest.pop(0) if ctx.have[0] else None
    seq_offsets, row_pos = (rest[0], rest[1]) if ctx.have[1] else (None, None)
    d = d_tok.contiguous()
    # the kernel sequence uses this buffer as scratch: in place only when the producer handed it over (ops._consumable)
    # (`d is d_tok`: contiguous() returned its argument -- no data_ptr() here, this also runs on fake tensors when traced)
    if d is d_tok and not getattr(d_tok, "_xfmr_consumable", False):
        d = d.clone()
    *sc, handles = ctx.scalars
    grads = torch.ops.xfmr.encoder_bwd(flat_params, d, key_mask, acts, *sc, step_device, handles, seq_offsets, row_pos,
                                       ctx.packed_seq_len)
    return (grads,) + (None,) * (n_in - 1)


encoder.register_autograd(_encoder_backward, setup_context=_encoder_setup)


# ------------------------------------------------------------------------------------------------ losses
def _loss_kw(train_head, all_heads, mask_false_negatives, mode, scale, margin, precision, num_hard_negatives):
    return dict(train_head=int(train_head), all_heads=int(all_heads), mask_false_negatives=bool(mask_false_negatives),
                mode=int(mode), scale=float(scale), margin=float(margin), precision=precision,
                num_hard_negatives=int(num_hard_negatives))


@torch.library.custom_op("xfmr::scale_by_device_scalar_", mutates_args=("x",))
def scale_by_device_scalar_(x: Tensor, g: Tensor) -> None:
    N.check(N.load().xfmr_scale_by_device_scalar(N.ptr(x), x.numel(), N.ptr(g), N.stream()), "xfmr_scale_by_device_scalar")


@scale_by_device_scalar_.register_fake
def _(x, g):
    return None


@torch.library.custom_op("xfmr::scale_by_device_scalar", mutates_args=())
def scale_by_device_scalar(x: Tensor, g: Tensor) -> Tensor:
    """x * g (g a device scalar) as a NEW tensor: what the registered autograd formulas use -- a backward that rescaled a
    forward output in place is not a graph AOTAutograd's partitioner accepts ("node was invalid, but is output")."""
    y = x.clone()
    N.check(N.load().xfmr_scale_by_device_scalar(N.ptr(y), y.numel(), N.ptr(g), N.stream()), "xfmr_scale_by_device_scalar")
    return y


@scale_by_device_scalar.register_fake
def _(x, g):
    return torch.empty_like(x)


@torch.library.custom_op("xfmr::sampled_loss", mutates_args=())
def sampled_loss(tok: Tensor, key_mask: Tensor, pos_idx: Tensor, neg_idx: Optional[Tensor], table: Tensor, rnorm: Tensor,
                 table_bf16: Optional[Tensor], train_head: int, all_heads: int, mask_false_negatives: bool, mode: int,
                 scale: float, margin: float, precision: str, num_hard_negatives: int, need_grad: bool,
                 plan: List[int]) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """(train_loss (), losses (14,), stats (16,), d_tok like tok -- or (0,) without ``need_grad``). The gradient
    dL(train_head)/dtok is produced in the same pass as the values (one launch sequence: xfmr_sampled_loss) and only
    scaled by the upstream gradient in backward. ``plan`` = [nsplit, nsplit_grad] (0 / [] = the library's launch plan).
    (An autograd-registered op must be functional: the eager step's extras -- a caller-zeroed gradient buffer, a
    workspace prepared on another stream, profile events -- exist on the ``autograd.Function`` route of :mod:`ops` only.)"""
    from . import ops

    losses, stats, d_tok = ops.sampled_loss(
        tok, key_mask, pos_idx, neg_idx, table, rnorm, table_bf16=table_bf16, need_grad=need_grad,
        nsplit=plan[0] if plan else 0, nsplit_grad=plan[1] if len(plan) > 1 else 0,
        **_loss_kw(train_head, all_heads, mask_false_negatives, mode, scale, margin, precision, num_hard_negatives))
    return losses[train_head].clone(), losses, stats, d_tok if d_tok is not None else tok.new_empty((0,))


def _loss_fake(like, grad_like, need_grad):
    return (like.new_empty((), dtype=torch.float32), like.new_empty((2 * N.NUM_LOSSES,), dtype=torch.float32),
            like.new_empty((N.NUM_STATS,), dtype=torch.float32),
            torch.empty_like(grad_like, dtype=torch.float32) if need_grad else like.new_empty((0,), dtype=torch.float32))


@sampled_loss.register_fake
def _(tok, key_mask, pos_idx, neg_idx, table, rnorm, table_bf16, train_head, all_heads, mask_false_negatives, mode, scale,
      margin, precision, num_hard_negatives, need_grad, plan):
    return _loss_fake(tok, tok, need_grad)


def _loss_setup(ctx, inputs, output):
    ctx.save_for_backward(output[3])
    ctx.n_in = len(inputs)
    ctx.set_materialize_grads(False)


def _make_loss_backward(hand_over: bool):
    def backward(ctx, g, _gl, _gs, _gd):
        (d,) = ctx.saved_tensors
        if g is None or d.numel() == 0:
            return (None,) * ctx.n_in
        d = torch.ops.xfmr.scale_by_device_scalar(d, g.contiguous().to(torch.float32))
        if hand_over and not torch.is_grad_enabled():  # a fresh buffer: the encoder backward may use it as scratch
            d._xfmr_consumable = True
        return (d,) + (None,) * (ctx.n_in - 1)

    return backward


sampled_loss.register_autograd(_make_loss_backward(True), setup_context=_loss_setup)


@torch.library.custom_op("xfmr::sampled_loss_lists", mutates_args=())
def sampled_loss_lists(query: Tensor, pos_items: Tensor, neg_items: Optional[Tensor], table: Tensor, rnorm: Tensor,
                       table_bf16: Optional[Tensor], train_head: int, all_heads: int, mask_false_negatives: bool,
                       mode: int, scale: float, margin: float, precision: str, num_hard_negatives: int,
                       need_grad: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """List form (compacted queries; losses.py:128-155 on structured candidates): (train_loss, losses, stats, d_query)."""
    from . import ops

    losses, stats, d_q = ops.sampled_loss_lists(
        query, pos_items, neg_items, table, rnorm, table_bf16=table_bf16, need_grad=need_grad,
        **_loss_kw(train_head, all_heads, mask_false_negatives, mode, scale, margin, precision, num_hard_negatives))
    return losses[train_head].clone(), losses, stats, d_q if d_q is not None else query.new_empty((0,))


@sampled_loss_lists.register_fake
def _(query, pos_items, neg_items, table, rnorm, table_bf16, train_head, all_heads, mask_false_negatives, mode, scale,
      margin, precision, num_hard_negatives, need_grad):
    return _loss_fake(query, query, need_grad)


sampled_loss_lists.register_autograd(_make_loss_backward(False), setup_context=_loss_setup)


@torch.library.custom_op("xfmr::dense_loss", mutates_args=())
def dense_loss(query: Tensor, cand: Tensor, target: Optional[Tensor], target_mode: int, train_head: int, all_heads: int,
               mask_false_negatives: bool, scale: float, margin: float, num_hard_negatives: int, need_query_grad: bool,
               need_cand_grad: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """``EmbedLoss.forward(query (N,H), candidates (N,C,H), target)`` exactly as the reference declares it
    (losses.py:128-155): (train_loss, losses, stats, d_query, d_cand); a gradient that is not wanted comes back as (0,).
    target_mode: XFMR_TARGET_*."""
    from . import ops

    tp = {N.TARGET_FIRST: "first", N.TARGET_DIAGONAL: "diagonal", N.TARGET_EXPLICIT: None}[int(target_mode)]
    out = ops.dense_loss(query, cand, target, target_position=tp, train_head=int(train_head), all_heads=int(all_heads),
                         mask_false_negatives=bool(mask_false_negatives), num_hard_negatives=int(num_hard_negatives),
                         scale=float(scale), margin=float(margin), need_grad=need_query_grad or need_cand_grad,
                         need_cand_grad=need_cand_grad)
    losses, stats, d_q = out[:3]
    d_c = out[3] if need_cand_grad else None
    empty = query.new_empty((0,))
    return (losses[train_head].clone(), losses, stats, d_q if (need_query_grad and d_q is not None) else empty,
            d_c if d_c is not None else empty)


@dense_loss.register_fake
def _(query, cand, target, target_mode, train_head, all_heads, mask_false_negatives, scale, margin, num_hard_negatives,
      need_query_grad, need_cand_grad):
    a = _loss_fake(query, query, need_query_grad)
    return a + (torch.empty_like(cand, dtype=torch.float32) if need_cand_grad else query.new_empty((0,), dtype=torch.float32),)


def _dense_setup(ctx, inputs, output):
    ctx.save_for_backward(output[3], output[4])
    ctx.n_in = len(inputs)
    ctx.set_materialize_grads(False)


def _dense_backward(ctx, g, _gl, _gs, _gq, _gc):
    if g is None:
        return (None,) * ctx.n_in
    g = g.contiguous().to(torch.float32)
    outs = []
    for t in ctx.saved_tensors:
        outs.append(torch.ops.xfmr.scale_by_device_scalar(t, g) if t.numel() else None)
    return (outs[0], outs[1]) + (None,) * (ctx.n_in - 2)


dense_loss.register_autograd(_dense_backward, setup_context=_dense_setup)


# ------------------------------------------------------------------------------------------------ small ops
@torch.library.custom_op("xfmr::l2_normalize", mutates_args=())
def l2_normalize(x: Tensor, eps: float) -> Tuple[Tensor, Tensor]:
    """(F.normalize(x, dim=-1), 1 / max(||x||, eps) per row)."""
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    y, inv = torch.empty_like(x), x.new_empty((rows,))
    N.check(N.load().xfmr_l2_normalize_fwd(N.ptr(x), N.ptr(y), N.ptr(inv), rows, H, eps, N.stream()), "xfmr_l2_normalize_fwd")
    return y, inv


@l2_normalize.register_fake
def _(x, eps):
    return torch.empty_like(x), x.new_empty((x.numel() // x.shape[-1],))


@torch.library.custom_op("xfmr::l2_normalize_bwd", mutates_args=())
def l2_normalize_bwd(dy: Tensor, y: Tensor, inv: Tensor, eps: float) -> Tensor:
    dx = torch.empty_like(y)
    rows, H = y.numel() // y.shape[-1], y.shape[-1]
    N.check(N.load().xfmr_l2_normalize_bwd(N.ptr(dy), N.ptr(y), N.ptr(inv), N.ptr(dx), rows, H, eps, N.stream()),
            "xfmr_l2_normalize_bwd")
    return dx


@l2_normalize_bwd.register_fake
def _(dy, y, inv, eps):
    return torch.empty_like(y)


def _l2_setup(ctx, inputs, output):
    ctx.save_for_backward(output[0], output[1])
    ctx.eps = inputs[1]


def _l2_backward(ctx, dy, _dinv):
    y, inv = ctx.saved_tensors
    return torch.ops.xfmr.l2_normalize_bwd(dy.contiguous().to(torch.float32), y, inv, ctx.eps), None


l2_normalize.register_autograd(_l2_backward, setup_context=_l2_setup)


@torch.library.custom_op("xfmr::pool", mutates_args=())
def pool(tok: Tensor, key_mask: Tensor, mode: int) -> Tensor:
    """sentence-transformers Pooling(pooling_mode) over (B,L,H) token embeddings; forward only. mode: mean 0, max 1, cls 2,
    lasttoken 3."""
    B, L, H = tok.shape
    out = tok.new_empty((B, H))
    N.check(N.load().xfmr_pool(N.ptr(tok), N.ptr(key_mask), N.ptr(out), B, L, H, int(mode), N.stream()), "xfmr_pool")
    return out


@pool.register_fake
def _(tok, key_mask, mode):
    return tok.new_empty((tok.shape[0], tok.shape[2]))


@torch.library.custom_op("xfmr::adamw_", mutates_args=("params", "exp_avg", "exp_avg_sq"))
def adamw_(params: Tensor, grads: Tensor, exp_avg: Tensor, exp_avg_sq: Tensor, lr: float, beta1: float, beta2: float,
           eps: float, weight_decay: float, step: int, grad_scale: float, step_device: Optional[Tensor]) -> None:
    """One AdamW step over the flat buffer (torch.optim.AdamW semantics); ``step_device``: read the step count on the device."""
    from . import ops

    ops.adamw_(params, grads, exp_avg, exp_avg_sq, lr=lr, beta1=beta1, beta2=beta2, eps=eps, weight_decay=weight_decay,
               step=step, grad_scale=grad_scale, step_device=step_device)


@adamw_.register_fake
def _(*a, **k):
    return None


OP_NAMES = ("encoder", "encoder_bwd", "sampled_loss", "sampled_loss_lists", "dense_loss", "scale_by_device_scalar_",
            "scale_by_device_scalar",
            "l2_normalize", "l2_normalize_bwd", "pool", "adamw_")
