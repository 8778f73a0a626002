"""Thin functional wrappers over the C ABI + the two autograd Functions of the training path.

Every function here enqueues HIP kernels from ``libxfmr_hip.so`` on the current stream of the current
device. Tensors are allocated with torch (device memory + caching allocator are plumbing); the
arithmetic is never torch's.
"""

from __future__ import annotations

import ctypes as C
import os

import torch

from . import _native as N

f32 = torch.float32


def _empty(shape, like: torch.Tensor, dtype=f32):
    return torch.empty(shape, dtype=dtype, device=like.device)


def _bytes(n: int, like: torch.Tensor):
    return torch.empty(max(int(n), 16), dtype=torch.uint8, device=like.device)


def selftest_mfma(device="cuda") -> list[int]:
    out = torch.full((4 + 4096,), -1, dtype=torch.int32, device=device)  # 4 result words + 16 KiB scratch
    N.check(N.load().xfmr_selftest_mfma(N.ptr(out), N.stream()), "xfmr_selftest_mfma")
    return out[:4].tolist()


# ------------------------------------------------------------------------------------------------ per-op
def embed_ln_fwd(item_idx, table, pos_emb, type_emb, gamma, beta, *, eps=1e-12, dropout_p=0.0, seed=0, site=0):
    B, L = item_idx.shape
    H = table.shape[1]
    out, pre = _empty((B, L, H), table), _empty((B, L, H), table)
    mean, rstd = _empty((B, L), table), _empty((B, L), table)
    mask = _empty((B, L), table, torch.uint8)
    N.check(
        N.load().xfmr_embed_ln_fwd(
            N.ptr(item_idx), N.ptr(table), table.shape[0], N.ptr(pos_emb), N.ptr(type_emb), N.ptr(gamma),
            N.ptr(beta), N.ptr(out), N.ptr(pre), N.ptr(mean), N.ptr(rstd), N.ptr(mask), B, L, H, eps, dropout_p,
            seed, site, N.stream(),
        ),
        "xfmr_embed_ln_fwd",
    )
    return out, pre, mean, rstd, mask


def embed_param_grads(d_pre, max_pos):
    B, L, H = d_pre.shape
    d_pos, d_type = _empty((max_pos, H), d_pre), _empty((2, H), d_pre)
    N.check(
        N.load().xfmr_embed_param_grads(N.ptr(d_pre), N.ptr(d_pos), N.ptr(d_type), B, L, H, max_pos, N.stream()),
        "xfmr_embed_param_grads",
    )
    return d_pos, d_type


def layernorm_fwd(x, gamma, beta, eps=1e-12):
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    y, mean, rstd = torch.empty_like(x), _empty((rows,), x), _empty((rows,), x)
    N.check(
        N.load().xfmr_layernorm_fwd(N.ptr(x), N.ptr(gamma), N.ptr(beta), N.ptr(y), N.ptr(mean), N.ptr(rstd), rows, H,
                                    eps, N.stream()),
        "xfmr_layernorm_fwd",
    )
    return y, mean, rstd


def layernorm_bwd(dy, x, mean, rstd, gamma, *, dropout_p=0.0, seed=0, site=0):
    rows, H = x.numel() // x.shape[-1], x.shape[-1]
    lib = N.load()
    dx = torch.empty_like(x)
    d_lin = torch.empty_like(x) if dropout_p > 0 else None
    dg, db, dbias = _empty((H,), x), _empty((H,), x), _empty((H,), x)
    ws = _bytes(lib.xfmr_layernorm_bwd_workspace(rows, H), x)
    N.check(
        lib.xfmr_layernorm_bwd(N.ptr(dy), N.ptr(x), N.ptr(mean), N.ptr(rstd), N.ptr(gamma), N.ptr(dx), N.ptr(d_lin),
                               N.ptr(dg), N.ptr(db), N.ptr(dbias), rows, H, dropout_p, seed, site, N.ptr(ws),
                               N.stream()),
        "xfmr_layernorm_bwd",
    )
    return dx, d_lin, dg, db, dbias


def linear_fwd(x, w, bias, *, epilogue=N.EPI_BIAS, residual=None, dropout_p=0.0, seed=0, site=0, precision="bf16"):
    M, K = x.numel() // x.shape[-1], x.shape[-1]
    Nout = w.shape[0]
    y = _empty((*x.shape[:-1], Nout), x)
    aux = torch.empty_like(y) if epilogue == N.EPI_BIAS_GELU else None
    N.check(
        N.load().xfmr_linear_fwd(N.ptr(x), N.ptr(w), N.ptr(bias), N.ptr(y), M, Nout, K, epilogue, N.ptr(residual),
                                 N.ptr(aux), dropout_p, seed, site, N.precision_id(precision), N.stream()),
        "xfmr_linear_fwd",
    )
    return (y, aux) if aux is not None else y


def linear_bwd_dx(dy, w, *, residual_grad=None, gelu_pre=None, precision="bf16"):
    M, Nout = dy.numel() // dy.shape[-1], dy.shape[-1]
    K = w.shape[1]
    dx = _empty((*dy.shape[:-1], K), dy)
    N.check(
        N.load().xfmr_linear_bwd_dx(N.ptr(dy), N.ptr(w), N.ptr(dx), M, Nout, K, N.ptr(residual_grad),
                                    N.ptr(gelu_pre), N.precision_id(precision), N.stream()),
        "xfmr_linear_bwd_dx",
    )
    return dx


def linear_bwd_dw(dy, x, *, precision="bf16"):
    M, Nout = dy.numel() // dy.shape[-1], dy.shape[-1]
    K = x.shape[-1]
    lib = N.load()
    dw = _empty((Nout, K), dy)
    nbytes = lib.xfmr_linear_bwd_dw_workspace(M, Nout, K)
    ws = _bytes(nbytes, dy)
    N.check(
        lib.xfmr_linear_bwd_dw(N.ptr(dy), N.ptr(x), N.ptr(dw), M, Nout, K, N.precision_id(precision), N.ptr(ws),
                               nbytes, N.stream()),
        "xfmr_linear_bwd_dw",
    )
    return dw


def colsum(a):
    M, Nc = a.numel() // a.shape[-1], a.shape[-1]
    lib = N.load()
    out = _empty((Nc,), a)
    ws = _bytes(lib.xfmr_colsum_workspace(M, Nc), a)
    N.check(lib.xfmr_colsum(N.ptr(a), N.ptr(out), M, Nc, N.ptr(ws), N.stream()), "xfmr_colsum")
    return out


def attn_fwd(qkv, key_mask, heads, *, dropout_p=0.0, seed=0, site=0, precision="bf16", causal=True):
    B, L, H3 = qkv.shape
    H = H3 // 3
    ctx, lse = _empty((B, L, H), qkv), _empty((B, heads, L), qkv)
    lib = N.load()
    head = (N.ptr(qkv), N.ptr(key_mask), N.ptr(ctx), N.ptr(lse), B, L, heads, H, dropout_p, seed, site,
            N.precision_id(precision))
    if causal:
        N.check(lib.xfmr_attn_fwd(*head, N.stream()), "xfmr_attn_fwd")
    else:
        N.check(lib.xfmr_attn_fwd_mode(*head, N.ATTN_BIDIRECTIONAL, N.stream()), "xfmr_attn_fwd_mode")
    return ctx, lse


def attn_bwd(qkv, key_mask, ctx, lse, d_ctx, heads, *, dropout_p=0.0, seed=0, site=0, precision="bf16", causal=True):
    B, L, H3 = qkv.shape
    d_qkv = torch.empty_like(qkv)
    lib = N.load()
    head = (N.ptr(qkv), N.ptr(key_mask), N.ptr(ctx), N.ptr(lse), N.ptr(d_ctx), N.ptr(d_qkv), B, L, heads, H3 // 3,
            dropout_p, seed, site, N.precision_id(precision))
    if causal:
        N.check(lib.xfmr_attn_bwd(*head, N.stream()), "xfmr_attn_bwd")
    else:
        N.check(lib.xfmr_attn_bwd_mode(*head, N.ATTN_BIDIRECTIONAL, N.stream()), "xfmr_attn_bwd_mode")
    return d_qkv


def table_rnorm(table):
    out = _empty((table.shape[0],), table)
    N.check(N.load().xfmr_table_rnorm(N.ptr(table), N.ptr(out), table.shape[0], table.shape[1], N.stream()),
            "xfmr_table_rnorm")
    return out


def table_prepare(table):
    """(rnorm, table_bf16): per-item inverse norms + the bf16 gather copy of the frozen table."""
    rn = _empty((table.shape[0],), table)
    tb = torch.empty(table.shape, dtype=torch.bfloat16, device=table.device)
    N.check(N.load().xfmr_table_prepare(N.ptr(table), N.ptr(rn), N.ptr(tb), table.shape[0], table.shape[1],
                                        N.stream()), "xfmr_table_prepare")
    return rn, tb


def mean_pool(tok, key_mask):
    B, L, H = tok.shape
    out = _empty((B, H), tok)
    N.check(N.load().xfmr_mean_pool(N.ptr(tok), N.ptr(key_mask), N.ptr(out), B, L, H, N.stream()), "xfmr_mean_pool")
    return out


def pool(tok, key_mask, mode: str = "mean"):
    """sentence-transformers Pooling(pooling_mode) over (B,L,H) tokens; forward only."""
    if use_torch_ops():
        return torch.ops.xfmr.pool(tok, key_mask, N.POOL_MODES[mode])
    B, L, H = tok.shape
    out = _empty((B, H), tok)
    N.check(N.load().xfmr_pool(N.ptr(tok), N.ptr(key_mask), N.ptr(out), B, L, H, N.POOL_MODES[mode], N.stream()),
            "xfmr_pool")
    return out


class L2NormalizeFunction(torch.autograd.Function):
    """``torch.nn.functional.normalize(x, dim=-1)`` (``models.py:393-394``) on (rows, H)."""

    @staticmethod
    def forward(ctx, x, eps=1e-12):
        x = x.contiguous().to(f32)
        rows, H = x.numel() // x.shape[-1], x.shape[-1]
        y, inv = torch.empty_like(x), _empty((rows,), x)
        N.check(N.load().xfmr_l2_normalize_fwd(N.ptr(x), N.ptr(y), N.ptr(inv), rows, H, eps, N.stream()),
                "xfmr_l2_normalize_fwd")
        ctx.save_for_backward(y, inv)
        ctx.eps = eps
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dy = dy.contiguous().to(f32)
        dx = torch.empty_like(y)
        rows, H = y.numel() // y.shape[-1], y.shape[-1]
        N.check(N.load().xfmr_l2_normalize_bwd(N.ptr(dy), N.ptr(y), N.ptr(inv), N.ptr(dx), rows, H, ctx.eps, N.stream()),
                "xfmr_l2_normalize_bwd")
        return dx, None


def l2_normalize(x, eps: float = 1e-12):
    if use_torch_ops():
        return torch.ops.xfmr.l2_normalize(x.contiguous().to(f32), float(eps))[0]
    return L2NormalizeFunction.apply(x, eps)


def adamw_(params, grads, exp_avg, exp_avg_sq, *, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.01, step=1,
           grad_scale=1.0, step_device=None):
    """``step_device`` (int32 device tensor = completed steps): the step count is read on the device (``step`` ignored)."""
    if step_device is not None:
        N.check(
            N.load().xfmr_adamw_dev(N.ptr(params), N.ptr(grads), N.ptr(exp_avg), N.ptr(exp_avg_sq), params.numel(), lr,
                                    beta1, beta2, eps, weight_decay, N.ptr(step_device), 1, grad_scale, N.stream()),
            "xfmr_adamw_dev",
        )
        return
    N.check(
        N.load().xfmr_adamw(N.ptr(params), N.ptr(grads), N.ptr(exp_avg), N.ptr(exp_avg_sq), params.numel(), lr, beta1,
                            beta2, eps, weight_decay, step, grad_scale, N.stream()),
        "xfmr_adamw",
    )


def step_advance_(step_device):
    N.check(N.load().xfmr_step_advance(N.ptr(step_device), N.stream()), "xfmr_step_advance")


def _plan_flags(nsplit: int = 0, nsplit_grad: int = 0) -> int:
    """XFMR_LOSS_NSPLIT(n) | XFMR_LOSS_NSPLIT_GRAD(n): launch-plan overrides inside xfmr_loss_cfg.flags (0 = library's plan)."""
    assert 0 <= nsplit < 256 and 0 <= nsplit_grad < 256
    return (nsplit << 8) | (nsplit_grad << 16)


def sampled_loss_workspace(like, T, H, n_rows, *, train_head, all_heads=True, mask_false_negatives=True, mode=N.NEG_SHARED,
                           scale=1.0, margin=0.5, precision="bf16", num_hard_negatives=0, nsplit=0, nsplit_grad=0, **_):
    """An (unprepared) workspace for sampled_loss(..., workspace=...) on `like`'s device."""
    cfg = _loss_cfg(train_head, all_heads, mask_false_negatives, mode, scale, margin, precision, num_hard_negatives,
                    flags=_plan_flags(nsplit, nsplit_grad))
    return _bytes(N.load().xfmr_sampled_loss_workspace_cfg(C.byref(cfg), T, H, n_rows), like)


def sampled_loss_prepare(ws, key_mask, pos_idx, neg_idx, rnorm, n_rows, H, *, train_head, all_heads=True,
                         mask_false_negatives=True, mode=N.NEG_SHARED, scale=1.0, margin=0.5, precision="bf16",
                         num_hard_negatives=0, nsplit=0, nsplit_grad=0, **_):
    """The index-only half of the loss (query compaction, multiplicities, distinct negatives) into `ws`, on the
    current stream: needs the key mask, not the token embeddings. Follow with sampled_loss(..., workspace=ws,
    prepared=True) with the same options."""
    cfg = _loss_cfg(train_head, all_heads, mask_false_negatives, mode, scale, margin, precision, num_hard_negatives,
                    flags=_plan_flags(nsplit, nsplit_grad))
    N.check(
        N.load().xfmr_sampled_loss_prepare(C.byref(cfg), N.ptr(key_mask), N.ptr(pos_idx), N.ptr(neg_idx), N.ptr(rnorm),
                                           n_rows, key_mask.numel(), H, N.ptr(ws), ws.numel(), N.stream()),
        "xfmr_sampled_loss_prepare",
    )


def sampled_loss(tok, key_mask, pos_idx, neg_idx, table, rnorm, *, train_head, all_heads=True,
                 mask_false_negatives=True, mode=N.NEG_SHARED, scale=1.0, margin=0.5, precision="bf16",
                 need_grad=True, table_bf16=None, num_hard_negatives=0, workspace=None, prepared=False, d_tok_zeroed=None,
                 profile_grad=None, profile_log=None, nsplit=0, nsplit_grad=0, padded_positions=0):
    """Returns (losses[7], stats[16], d_tok or None). tok: (T,H) or (B,L,H). `workspace` (sampled_loss_workspace) +
    `prepared=True`: sampled_loss_prepare already ran on it; `d_tok_zeroed`: a zero-filled buffer like tok to take the
    gradient (the call then skips its own memset); `profile_grad` / `profile_log`: (start, stop) hipEvent_t handles
    recorded around the gradient-pass / logging-pass kernel (measurement); `nsplit` / `nsplit_grad`: column-split counts
    of the launch plan forced by the caller (0 = the library's plan; parity tests walk several plans)."""
    H = tok.shape[-1]
    T = tok.numel() // H
    lib = N.load()
    d_tok = (d_tok_zeroed if d_tok_zeroed is not None else torch.empty_like(tok)) if need_grad else None
    flags = _plan_flags(nsplit, nsplit_grad)
    if d_tok_zeroed is not None and need_grad:
        assert d_tok.shape == tok.shape and d_tok.dtype == tok.dtype
        flags |= N.LOSS_DTOK_ZEROED
    cfg = _loss_cfg(train_head, all_heads, mask_false_negatives, mode, scale, margin, precision, num_hard_negatives,
                    flags=flags, profile_grad=profile_grad, profile_log=profile_log, padded_positions=padded_positions)
    n_rows = table.shape[0]
    losses = _empty((2 * N.NUM_LOSSES,), tok)
    stats = _empty((N.NUM_STATS,), tok)
    if workspace is None:
        assert not prepared
        nbytes = lib.xfmr_sampled_loss_workspace_cfg(C.byref(cfg), T, H, n_rows)
        ws = _bytes(nbytes, tok)
    else:
        ws, nbytes = workspace, workspace.numel()
    fn = lib.xfmr_sampled_loss_prepared if prepared else lib.xfmr_sampled_loss
    N.check(
        fn(C.byref(cfg), N.ptr(tok), N.ptr(key_mask), N.ptr(pos_idx), N.ptr(neg_idx), N.ptr(table), N.ptr(rnorm),
           N.ptr(table_bf16), n_rows, T, H, N.ptr(losses), N.ptr(stats), N.ptr(d_tok), N.ptr(ws), nbytes, N.stream()),
        "xfmr_sampled_loss_prepared" if prepared else "xfmr_sampled_loss",
    )
    return losses, stats, d_tok


def _loss_cfg(train_head, all_heads, mask_false_negatives, mode, scale, margin, precision,
              num_hard_negatives=0, *, flags=0, profile_grad=None, profile_log=None, padded_positions=0) -> N.LossCfg:
    cfg = N.LossCfg(
        train_head=N.LOSS_IDS[train_head] if isinstance(train_head, str) else int(train_head),
        all_heads=int(all_heads), mask_false_negatives=int(mask_false_negatives), mode=mode,
        precision=N.precision_id(precision), scale=float(scale), margin=float(margin),
        num_hard_negatives=int(num_hard_negatives), flags=int(flags), padded_positions=int(padded_positions),
    )
    if profile_grad is not None:  # (start, stop) hipEvent_t handles around the gradient-pass kernel
        cfg.profile_grad[0], cfg.profile_grad[1] = profile_grad
    if profile_log is not None:  # ... around the values-only logging pass
        cfg.profile_log[0], cfg.profile_log[1] = profile_log
    return cfg


def dense_loss(query, cand, target=None, *, target_position="first", train_head, all_heads=False,
               mask_false_negatives=True, num_hard_negatives=0, scale=1.0, margin=0.5, need_grad=True,
               need_cand_grad=False):
    """``EmbedLoss.forward`` on a dense (N,C,H) candidate tensor (``losses.py:128-155``):
    returns (losses[14], stats[16], d_query or None[, d_cand when ``need_cand_grad``])."""
    Nq, H = query.shape
    Cn = cand.shape[1]
    mode = {"first": N.TARGET_FIRST, "diagonal": N.TARGET_DIAGONAL, None: N.TARGET_EXPLICIT}[target_position]
    lib = N.load()
    cfg = _loss_cfg(train_head, all_heads, mask_false_negatives, N.NEG_SHARED, scale, margin, "fp32",
                    num_hard_negatives)
    losses, stats = _empty((2 * N.NUM_LOSSES,), query), _empty((N.NUM_STATS,), query)
    d_q = torch.empty_like(query) if need_grad else None
    nbytes = lib.xfmr_dense_loss_workspace(Nq, Cn, H)
    ws = _bytes(nbytes, query)
    d_c = torch.empty_like(cand) if need_cand_grad else None
    N.check(
        lib.xfmr_dense_loss_grads(C.byref(cfg), N.ptr(query), N.ptr(cand), N.ptr(target), mode, Nq, Cn, H, N.ptr(losses),
                                  N.ptr(stats), N.ptr(d_q), N.ptr(d_c), N.ptr(ws), nbytes, N.stream()),
        "xfmr_dense_loss_grads",
    )
    return (losses, stats, d_q, d_c) if need_cand_grad else (losses, stats, d_q)


def sampled_loss_lists(query, pos_items, neg_items, table, rnorm, *, train_head, all_heads=False,
                       mask_false_negatives=True, mode=N.NEG_SHARED, scale=1.0, margin=0.5, precision="bf16",
                       need_grad=True, table_bf16=None, num_hard_negatives=0):
    """List form (compacted queries): returns (losses[7], stats[16], d_query or None)."""
    Np, H = query.shape
    Nn = 0 if neg_items is None else neg_items.numel()
    lib = N.load()
    cfg = _loss_cfg(train_head, all_heads, mask_false_negatives, mode, scale, margin, precision, num_hard_negatives)
    n_rows = table.shape[0]
    losses, stats = _empty((2 * N.NUM_LOSSES,), query), _empty((N.NUM_STATS,), query)
    d_q = torch.empty_like(query) if need_grad else None
    nbytes = lib.xfmr_sampled_loss_lists_workspace_cfg(C.byref(cfg), Np, Nn, H, n_rows)
    ws = _bytes(nbytes, query)
    N.check(
        lib.xfmr_sampled_loss_lists(C.byref(cfg), N.ptr(query), N.ptr(pos_items), N.ptr(neg_items), Np, Nn,
                                    N.ptr(table), N.ptr(rnorm), N.ptr(table_bf16), n_rows, H, N.ptr(losses),
                                    N.ptr(stats), N.ptr(d_q),
                                    N.ptr(ws), nbytes, N.stream()),
        "xfmr_sampled_loss_lists",
    )
    return losses, stats, d_q


# ------------------------------------------------------------------------------------------------ encoder
def _env_on(name: str) -> bool:
    v = os.environ.get(name, "")
    return bool(v) and v != "0"


def encoder_flags_from_env() -> int:
    """A/B switches -> xfmr_encoder_cfg.flags bits. Read when the cfg of a step is made, so the
    forward and the backward of that step (which share the cfg) agree whatever happens to the environment in between."""
    f = 0
    if _env_on("XFMR_LN_UNFUSED"):
        f |= N.ENC_LN_UNFUSED
    if _env_on("XFMR_FFN_UNFUSED"):
        f |= N.ENC_FFN_UNFUSED
    if _env_on("XFMR_FFN_BWD_UNFUSED"):
        f |= N.ENC_FFN_BWD_UNFUSED
    if os.environ.get("XFMR_DW_SIDE", "") == "0":
        f |= N.ENC_DW_INLINE
    if os.environ.get("XFMR_DW_SIDE", "") == "any":  # (experiments: the side stream below its token threshold too)
        f |= N.ENC_DW_SIDE_ANY
    if os.environ.get("XFMR_DW_PAIR", "") == "0":
        f |= N.ENC_DW_UNPAIRED
    if _env_on("XFMR_REDUCE_HALF_EARLY"):
        f |= N.ENC_REDUCE_HALF_EARLY
    return f


def make_encoder_cfg(*, batch, seq_len, hidden, heads, inter, layers, max_pos, precision, ln_eps=1e-12,
                     hidden_dropout=0.0, attn_dropout=0.0, seed=0, causal=True, flags=None, step_device=None,
                     embed_event=None, context=None, grads_half_event=None, extra_flags=0, profile=None,
                     seq_offsets=None, row_pos=None) -> N.EncoderCfg:
    """``step_device``: a uint32 device tensor (or pointer) mixed into the dropout stream on the device;
    ``embed_event``: a hipEvent_t handle the forward records once the key mask exists; ``context``: an
    ``xfmr_context`` handle (side stream of the backward's weight-gradient GEMMs); ``profile``: (N.PROF_*, layer,
    event0, event1) -- HIP events recorded around that part of the encoder (measurement); ``seq_offsets`` (batch + 1,
    int32) + ``row_pos`` (packed rows, int32): the PACKED layout (pack_rows) -- the token axis holds only each
    sequence's own rows."""
    f = encoder_flags_from_env() if flags is None else int(flags)
    f |= int(extra_flags)
    if not causal:
        f |= N.ENC_BIDIRECTIONAL
    if isinstance(step_device, torch.Tensor):
        assert step_device.dtype in (torch.int32, torch.uint32) and step_device.is_cuda
        step_device = step_device.data_ptr()
    return N.EncoderCfg(
        batch=batch, seq_len=seq_len, hidden=hidden, heads=heads, inter=inter, layers=layers, max_pos=max_pos,
        precision=N.precision_id(precision), ln_eps=ln_eps, hidden_dropout=hidden_dropout,
        attn_dropout=attn_dropout, flags=f, seed=seed, step_device=step_device, embed_event=embed_event,
        context=context, grads_half_event=grads_half_event,
        profile_kernel=int(profile[0]) if profile else 0, profile_layer=int(profile[1]) if profile else 0,
        profile_events=(C.c_void_p * 2)(profile[2], profile[3]) if profile else (C.c_void_p * 2)(None, None),
        seq_offsets=N.ptr(seq_offsets), row_pos=N.ptr(row_pos), packed_rows=0 if row_pos is None else int(row_pos.numel()),
    )


class Context:
    """Owner of one ``xfmr_context`` (the lowest-priority side stream + fork / join events that ``xfmr_encoder_bwd`` runs
    its weight-gradient GEMMs on). Created on the current device; destroyed with the object."""

    def __init__(self, device=None):
        self.handle = None
        h = C.c_void_p()
        with torch.cuda.device(device):
            N.check(N.load().xfmr_context_create(C.byref(h)), "xfmr_context_create")
        self.handle = h.value

    def close(self):
        if self.handle:
            N.load().xfmr_context_destroy(self.handle)
            self.handle = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


def pack_rows(hist, pos, neg, offsets64, rows: int, order=None):
    """The PACKED layout of a collated batch (xfmr_pack_rows): hist / pos / neg (B, L) int64 device tensors (neg may be
    None), offsets64 (B + 1) int64 on the device = the cumulative row lengths, ``rows`` = offsets64[-1] as the HOST knows
    it (the collate padded the rows: it knows their lengths). ``order`` (B,) int64 on the device: packed slot b holds batch row
    order[b] and offsets64 are the cumulative lengths in THAT order (:func:`length_order`: longest first). Returns the packed
    (rows,) index tensors, ``seq_offsets`` (B + 1) int32 and ``row_pos`` (rows,) int32 for make_encoder_cfg."""
    B, L = hist.shape
    dev = hist.device
    out = {k: torch.empty((rows,), dtype=torch.int64, device=dev) for k in ("hist", "pos")}
    out["neg"] = torch.empty((rows,), dtype=torch.int64, device=dev) if neg is not None else None
    out["seq_offsets"] = torch.empty((B + 1,), dtype=torch.int32, device=dev)
    out["row_pos"] = torch.empty((rows,), dtype=torch.int32, device=dev)
    N.check(N.load().xfmr_pack_rows_ordered(N.ptr(hist), N.ptr(pos), N.ptr(neg), N.ptr(offsets64), N.ptr(order), B, L, rows,
                                            N.ptr(out["hist"]), N.ptr(out["pos"]), N.ptr(out["neg"]), N.ptr(out["seq_offsets"]),
                                            N.ptr(out["row_pos"]), N.stream()), "xfmr_pack_rows_ordered")
    return out


def length_order(lengths):
    """(order, offsets) of a batch's row lengths (CPU int64 tensor) for the packed layout: the rows by length, LONGEST first
    (stable), and the cumulative lengths in that order (B + 1). The attention kernels run one workgroup per (sequence, head)
    in slot order; a 200-token sequence that starts last is a 28 us tail on a launch whose balanced time is ~45 (MovieLens-like
    batches of 512, one-stream trace: attention backward 72.6 -> 58.2 us per layer, forward 26.2 -> 20.2; step 1.867 -> 1.853 ms)."""
    lens = torch.as_tensor(lengths, dtype=torch.int64).cpu()
    if os.environ.get("XFMR_PACK_ORDER", "") == "0":  # (A/B runs: the batch's own order)
        order = torch.arange(lens.numel(), dtype=torch.int64)
    else:
        order = torch.argsort(lens, descending=True, stable=True)
    off = torch.zeros(lens.numel() + 1, dtype=torch.int64)
    torch.cumsum(lens[order], 0, out=off[1:])
    return order, off


def encoder_fwd(cfg: N.EncoderCfg, flat_params, item_idx, table):
    lib = N.load()
    T, H = cfg.batch * cfg.seq_len, cfg.hidden
    if cfg.packed_rows:  # packed layout: the token axis holds the sequences' own rows only
        tok = _empty((cfg.packed_rows, H), flat_params)
        key_mask = _empty((cfg.packed_rows,), flat_params, torch.uint8)
    else:
        tok = _empty((cfg.batch, cfg.seq_len, H), flat_params)
        key_mask = _empty((cfg.batch, cfg.seq_len), flat_params, torch.uint8)
    nbytes = lib.xfmr_encoder_workspace_bytes(C.byref(cfg))
    if nbytes == 0:
        raise RuntimeError("xfmr_encoder_workspace_bytes: unsupported encoder configuration "
                           "(head size must be 32, sizes positive, seq_len <= max_pos)")
    acts = _bytes(nbytes, flat_params)
    N.check(
        lib.xfmr_encoder_fwd(C.byref(cfg), N.ptr(flat_params), N.ptr(item_idx), N.ptr(table), table.shape[0],
                             N.ptr(tok), N.ptr(key_mask), N.ptr(acts), nbytes, N.stream()),
        "xfmr_encoder_fwd",
    )
    return tok, key_mask, acts


last_encoder_grad_ptr = 0  # device address of the flat gradient buffer the last xfmr_encoder_bwd call wrote


def encoder_bwd(cfg: N.EncoderCfg, flat_params, d_tok, key_mask, acts, grads=None):
    """d_tok is clobbered. Returns the flat gradient buffer."""
    global last_encoder_grad_ptr
    if grads is None:
        grads = torch.empty_like(flat_params)
    last_encoder_grad_ptr = grads.data_ptr()  # (distributed.HalvedAllReduce checks that .grad IS this buffer)
    N.check(
        N.load().xfmr_encoder_bwd(C.byref(cfg), N.ptr(flat_params), N.ptr(grads), N.ptr(d_tok), N.ptr(key_mask),
                                  N.ptr(acts), acts.numel(), N.stream()),
        "xfmr_encoder_bwd",
    )
    return grads


class EncoderFunction(torch.autograd.Function):
    """tok, key_mask = encoder(flat_params, item_idx); backward fills the flat gradient."""

    @staticmethod
    def forward(ctx, flat_params, item_idx, table, cfg, keep=None):
        tok, key_mask, acts = encoder_fwd(cfg, flat_params, item_idx, table)
        ctx.cfg = cfg
        ctx.keep = keep  # tensors the cfg points into (packed layout: seq_offsets, row_pos) live until the backward has run
        ctx.save_for_backward(flat_params, key_mask, acts)
        ctx.mark_non_differentiable(key_mask)
        ctx.set_materialize_grads(False)  # no zero-filled gradient for the mask output
        return tok, key_mask

    @staticmethod
    def backward(ctx, d_tok, _d_mask):
        flat_params, key_mask, acts = ctx.saved_tensors
        if d_tok is None:
            return None, None, None, None, None
        d = d_tok.contiguous()
        # the kernel sequence reuses this buffer as scratch: it may only do so in place when the producer hands the
        # buffer over (the fused loss does: its saved gradient is dead after its own backward)
        if d.data_ptr() == d_tok.data_ptr() and not getattr(d_tok, "_xfmr_consumable", False):
            d = d.clone()
        grads = encoder_bwd(ctx.cfg, flat_params, d, key_mask, acts)
        return grads, None, None, None, None


_UNIT_GRADS: dict = {}


def unit_grad(like: torch.Tensor) -> torch.Tensor:
    """THE scalar 1 on `like`'s device, one persistent tensor per device. ``loss.backward(gradient=ops.unit_grad(loss))``
    does what ``loss.backward()`` does without the per-step ones-fill launch, and the fused loss's backward recognises
    the object and skips its `d_tok *= g` launch (two ~5 us launches between the loss and the encoder backward).
    ``RecommenderLightningModule.backward`` -- Lightning's overridable hook -- passes it."""
    key = (like.device.type, like.device.index)
    t = _UNIT_GRADS.get(key)
    if t is None:
        t = _UNIT_GRADS[key] = torch.ones((), dtype=f32, device=like.device)
    return t


unit_grad_hits = 0  # backward calls that recognised the unit gradient (tests)


def _is_unit_grad(g: torch.Tensor) -> bool:
    global unit_grad_hits
    hit = any(g is t for t in _UNIT_GRADS.values())
    unit_grad_hits += int(hit)
    return hit


class SampledLossFunction(torch.autograd.Function):
    """(train_loss, losses[7], stats[16]) = fused_loss(tok, ...). Only train_loss carries a gradient; it was
    computed in the same pass as the forward and is scaled by the incoming gradient in backward."""

    @staticmethod
    def forward(ctx, tok, key_mask, pos_idx, neg_idx, table, rnorm, opts):
        need = tok.requires_grad
        losses, stats, d_tok = sampled_loss(tok, key_mask, pos_idx, neg_idx, table, rnorm, need_grad=need, **opts)
        head = opts["train_head"]
        head = N.LOSS_IDS[head] if isinstance(head, str) else head
        if need:
            ctx.save_for_backward(d_tok)
        ctx.mark_non_differentiable(losses, stats)
        ctx.set_materialize_grads(False)  # no zero-filled gradients for the logging outputs
        return losses[head].clone(), losses, stats

    @staticmethod
    def backward(ctx, g, _gl, _gs):
        if g is None:
            return (None,) * 7
        (d_tok,) = ctx.saved_tensors
        if not _is_unit_grad(g):  # (the persistent unit gradient: nothing to multiply by)
            g = g.contiguous().to(f32)
            N.check(N.load().xfmr_scale_by_device_scalar(N.ptr(d_tok), d_tok.numel(), N.ptr(g), N.stream()),
                    "xfmr_scale_by_device_scalar")
        if not torch.is_grad_enabled():  # not under create_graph: the buffer is dead after this backward
            d_tok._xfmr_consumable = True
        return d_tok, None, None, None, None, None, None


class SampledLossListsFunction(torch.autograd.Function):
    """List form of :class:`SampledLossFunction`: compacted queries, explicit item-id lists."""

    @staticmethod
    def forward(ctx, query, pos_items, neg_items, table, rnorm, opts):
        need = query.requires_grad
        losses, stats, d_q = sampled_loss_lists(query, pos_items, neg_items, table, rnorm, need_grad=need, **opts)
        head = opts["train_head"]
        head = N.LOSS_IDS[head] if isinstance(head, str) else head
        if need:
            ctx.save_for_backward(d_q)
        ctx.mark_non_differentiable(losses, stats)
        return losses[head].clone(), losses, stats

    @staticmethod
    def backward(ctx, g, _gl, _gs):
        (d_q,) = ctx.saved_tensors
        g = g.contiguous().to(f32)
        N.check(N.load().xfmr_scale_by_device_scalar(N.ptr(d_q), d_q.numel(), N.ptr(g), N.stream()),
                "xfmr_scale_by_device_scalar")
        return d_q, None, None, None, None, None


class DenseLossFunction(torch.autograd.Function):
    """``EmbedLoss.forward`` on dense candidates, differentiable in the query and -- when they require a gradient -- in
    the candidates (``losses.py:128-155``)."""

    @staticmethod
    def forward(ctx, query, cand, target, opts):
        need, need_c = query.requires_grad, cand.requires_grad
        out = dense_loss(query, cand, target, need_grad=need or need_c, need_cand_grad=need_c, **opts)
        losses, stats, d_q = out[:3]
        head = opts["train_head"]
        head = N.LOSS_IDS[head] if isinstance(head, str) else head
        ctx.flags = (need, need_c)
        ctx.save_for_backward(*([d_q] if need else []), *([out[3]] if need_c else []))
        ctx.mark_non_differentiable(losses, stats)
        return losses[head].clone(), losses, stats

    @staticmethod
    def backward(ctx, g, _gl, _gs):
        need, need_c = ctx.flags
        saved = list(ctx.saved_tensors)
        g = g.contiguous().to(f32)
        outs = []
        for t in saved:
            N.check(N.load().xfmr_scale_by_device_scalar(N.ptr(t), t.numel(), N.ptr(g), N.stream()),
                    "xfmr_scale_by_device_scalar")
            outs.append(t)
        d_q = outs.pop(0) if need else None
        d_c = outs.pop(0) if need_c else None
        return d_q, d_c, None, None


# ------------------------------------------------------------------------------------------------ torch.ops.xfmr.*
# The same implementations registered as PyTorch custom operators (custom_ops.py: schema + fake tensors + autograd
# formulas), so that torch.compile / torch.export of a model built from these kernels sees shape-inferable opaque ops.
# The eager step keeps the autograd.Function route above -- a Python-registered op with an autograd formula costs
# ~130 us of host time per call against ~12 (custom_ops.py) --; traced code, or XFMR_TORCH_OPS=1, takes the ops.
from . import custom_ops  # noqa: E402  (registers torch.ops.xfmr.*)


def use_torch_ops() -> bool:
    return torch.compiler.is_compiling() or os.environ.get("XFMR_TORCH_OPS", "") == "1"


def encoder_op_args(*, heads, inter, layers, max_pos, precision, ln_eps=1e-12, hidden_dropout=0.0, attn_dropout=0.0,
                    seed=0, causal=True, flags=None, step_device=None, embed_event=None, context=None,
                    grads_half_event=None, extra_flags=0, profile=None, seq_offsets=None, row_pos=None, seq_len=0, **_shape):
    """make_encoder_cfg's keyword arguments -> the scalar arguments of torch.ops.xfmr.encoder, in plain Python (no ctypes:
    this runs inside traced code). Returns (scalars..., step_device, handles)."""
    f = encoder_flags_from_env() if flags is None else int(flags)
    f |= int(extra_flags)
    if not causal:
        f |= N.ENC_BIDIRECTIONAL
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    if seed >= 1 << 63:
        seed -= 1 << 64
    prof = list(profile) if profile else [0, 0, 0, 0]
    handles = [int(embed_event or 0), int(context or 0), int(grads_half_event or 0)] + [int(x or 0) for x in prof]
    return (int(heads), int(inter), int(layers), int(max_pos), str(precision), float(ln_eps), float(hidden_dropout),
            float(attn_dropout), f, seed, step_device, handles, seq_offsets, row_pos,
            int(seq_len) if seq_offsets is not None else 0)


def encoder(flat_params, item_idx, table, cfg_kwargs: dict):
    """tok (B,L,H), key_mask (B,L) = the encoder of `cfg_kwargs` (make_encoder_cfg's keyword arguments) on `item_idx`;
    differentiable in `flat_params`. Eager: EncoderFunction; traced / XFMR_TORCH_OPS=1: torch.ops.xfmr.encoder."""
    if use_torch_ops():
        tok, key_mask, _acts = torch.ops.xfmr.encoder(flat_params, item_idx, table, *encoder_op_args(**cfg_kwargs))
        return tok, key_mask
    keep = (cfg_kwargs.get("seq_offsets"), cfg_kwargs.get("row_pos"))
    return EncoderFunction.apply(flat_params, item_idx, table, make_encoder_cfg(**cfg_kwargs), keep)


_EAGER_ONLY_LOSS_OPTS = ("workspace", "prepared", "d_tok_zeroed", "profile_grad", "profile_log")


def _loss_op_scalars(opts: dict):
    head = opts["train_head"]
    return (N.LOSS_IDS[head] if isinstance(head, str) else int(head), int(opts.get("all_heads", True)),
            bool(opts.get("mask_false_negatives", True)), int(opts.get("mode", N.NEG_SHARED)), float(opts.get("scale", 1.0)),
            float(opts.get("margin", 0.5)), str(opts.get("precision", "bf16")), int(opts.get("num_hard_negatives", 0)))


def sampled_loss_train(tok, key_mask, pos_idx, neg_idx, table, rnorm, opts: dict):
    """(train_loss, losses[14], stats[16]) on (B*L) positions; train_loss carries the gradient w.r.t. `tok`."""
    if use_torch_ops() and not any(opts.get(k) for k in _EAGER_ONLY_LOSS_OPTS):
        out = torch.ops.xfmr.sampled_loss(tok, key_mask, pos_idx, neg_idx, table, rnorm, opts.get("table_bf16"),
                                          *_loss_op_scalars(opts), bool(tok.requires_grad and torch.is_grad_enabled()),
                                          [int(opts.get("nsplit", 0)), int(opts.get("nsplit_grad", 0))])
        return out[0], out[1], out[2]
    return SampledLossFunction.apply(tok, key_mask, pos_idx, neg_idx, table, rnorm, opts)


def sampled_loss_lists_train(query, pos_items, neg_items, table, rnorm, opts: dict):
    if use_torch_ops():
        out = torch.ops.xfmr.sampled_loss_lists(query, pos_items, neg_items, table, rnorm, opts.get("table_bf16"),
                                                *_loss_op_scalars(opts), bool(query.requires_grad and torch.is_grad_enabled()))
        return out[0], out[1], out[2]
    return SampledLossListsFunction.apply(query, pos_items, neg_items, table, rnorm, opts)


def dense_loss_train(query, cand, target, opts: dict):
    if use_torch_ops():
        mode = {"first": N.TARGET_FIRST, "diagonal": N.TARGET_DIAGONAL, None: N.TARGET_EXPLICIT}[opts.get("target_position", "first")]
        head = opts["train_head"]
        grad_on = torch.is_grad_enabled()
        out = torch.ops.xfmr.dense_loss(
            query, cand, target, mode, N.LOSS_IDS[head] if isinstance(head, str) else int(head), int(opts.get("all_heads", False)),
            bool(opts.get("mask_false_negatives", True)), float(opts.get("scale", 1.0)), float(opts.get("margin", 0.5)),
            int(opts.get("num_hard_negatives", 0)), bool(query.requires_grad and grad_on), bool(cand.requires_grad and grad_on))
        return out[0], out[1], out[2]
    return DenseLossFunction.apply(query, cand, target, opts)
