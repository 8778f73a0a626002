"""Oracle: causal BERT encoder over ``inputs_embeds`` (TEST INFRASTRUCTURE).

The reference builds a stock HuggingFace ``BertModel(is_decoder=True)``
(``xfmr_rec/models.py:51-102``) and feeds it ``inputs_embeds`` +
``attention_mask`` (``models.py:343-345``). The arithmetic therefore lives in
the third-party ``transformers`` package (pinned 4.57.6 in the reference's
``uv.lock:3792-3793``; 5.15.0 in this container). It is restated here as
plain functions over an HF-keyed parameter dict, following
``TF:models/bert/modeling_bert.py``:

* embeddings  ``:68-108``   x + type_emb[0] + pos_emb[0:L] -> LayerNorm -> dropout
* attention   ``:111-203``  softmax(QK^T/sqrt(dh) + M) V, M = 0 where
  (k <= q and key_mask[b,k]) else finfo.min  (``TF:masking_utils.py:76-80,168-179``)
* self-output ``:282-293``  LN(dropout(dense(ctx)) + x)
* FFN         ``:325-351``  LN(dropout(dense2(gelu_erf(dense1(x)))) + x)

Dropout is applied only when ``dropout_p > 0`` (training-mode parity is not
bit-reproducible; parity tests run with dropout off, as eval()).
Sentence pooling follows sentence-transformers ``Pooling(mean)`` as wired at
``models.py:143-145``: sum(tok*m)/clamp(sum(m), 1e-9).
"""

from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LN_EPS = 1e-12  # TF:models/bert/configuration_bert.py layer_norm_eps default


def init_params(
    hidden_size: int,
    num_hidden_layers: int,
    intermediate_size: int,
    max_seq_length: int,
    *,
    seed: int = 0,
    dtype=torch.float32,
) -> dict[str, torch.Tensor]:
    """HF-keyed trainable tensors with HF BERT init (N(0, 0.02), LN = 1/0, bias 0).

    Only the tensors that receive a gradient on the training path are created
    (no ``word_embeddings`` / ``pooler``: SURVEY F12).
    """
    g = torch.Generator().manual_seed(seed)
    H, I = hidden_size, intermediate_size

    def w(*shape):
        return torch.randn(*shape, generator=g, dtype=dtype) * 0.02

    p = {
        "embeddings.position_embeddings.weight": w(max_seq_length, H),
        "embeddings.token_type_embeddings.weight": w(2, H),
        "embeddings.LayerNorm.weight": torch.ones(H, dtype=dtype),
        "embeddings.LayerNorm.bias": torch.zeros(H, dtype=dtype),
    }
    for i in range(num_hidden_layers):
        pre = f"encoder.layer.{i}."
        for name in ("query", "key", "value"):
            p[pre + f"attention.self.{name}.weight"] = w(H, H)
            p[pre + f"attention.self.{name}.bias"] = torch.zeros(H, dtype=dtype)
        p[pre + "attention.output.dense.weight"] = w(H, H)
        p[pre + "attention.output.dense.bias"] = torch.zeros(H, dtype=dtype)
        p[pre + "attention.output.LayerNorm.weight"] = torch.ones(H, dtype=dtype)
        p[pre + "attention.output.LayerNorm.bias"] = torch.zeros(H, dtype=dtype)
        p[pre + "intermediate.dense.weight"] = w(I, H)
        p[pre + "intermediate.dense.bias"] = torch.zeros(I, dtype=dtype)
        p[pre + "output.dense.weight"] = w(H, I)
        p[pre + "output.dense.bias"] = torch.zeros(H, dtype=dtype)
        p[pre + "output.LayerNorm.weight"] = torch.ones(H, dtype=dtype)
        p[pre + "output.LayerNorm.bias"] = torch.zeros(H, dtype=dtype)
    return p


def _dropout(x, p, training):
    return F.dropout(x, p=p, training=training) if (training and p > 0) else x


def embeddings_forward(p, inputs_embeds, *, dropout_p=0.0, training=False):
    """TF:models/bert/modeling_bert.py:68-108 with token_type_ids = 0, position_ids = arange(L)."""
    L = inputs_embeds.size(1)
    x = inputs_embeds + p["embeddings.token_type_embeddings.weight"][0]
    x = x + p["embeddings.position_embeddings.weight"][:L]
    x = F.layer_norm(
        x, (x.size(-1),), p["embeddings.LayerNorm.weight"], p["embeddings.LayerNorm.bias"], LN_EPS
    )
    return _dropout(x, dropout_p, training)


def causal_padding_bias(key_mask: torch.Tensor, dtype, causal: bool = True) -> torch.Tensor:
    """(B,L) key mask -> additive (B,1,L,L) bias. TF:masking_utils.py:76-80,168-179.

    ``causal=False`` is ``BertConfig(is_decoder=False)`` (``ModelConfig.is_decoder``, models.py:50,355): HF then
    builds the bidirectional mask, i.e. the key-padding condition alone."""
    B, L = key_mask.shape
    tri = torch.ones(L, L, dtype=torch.bool, device=key_mask.device)
    if causal:
        tri = tri.tril()
    allowed = tri[None, :, :] & key_mask.bool()[:, None, :]
    bias = torch.zeros(B, L, L, dtype=dtype, device=key_mask.device)
    bias.masked_fill_(~allowed, torch.finfo(dtype).min)
    return bias[:, None]


def self_attention(p, pre, x, bias, num_heads, *, dropout_p=0.0, training=False):
    """TF:models/bert/modeling_bert.py:111-203 (eager_attention_forward :111-136)."""
    B, L, H = x.shape
    dh = H // num_heads

    def proj(name):
        y = F.linear(x, p[pre + f"attention.self.{name}.weight"], p[pre + f"attention.self.{name}.bias"])
        return y.view(B, L, num_heads, dh).transpose(1, 2)

    q, k, v = proj("query"), proj("key"), proj("value")
    scores = torch.matmul(q, k.transpose(2, 3)) * dh**-0.5 + bias
    probs = _dropout(F.softmax(scores, dim=-1), dropout_p, training)
    ctx = torch.matmul(probs, v).transpose(1, 2).reshape(B, L, H)
    return ctx


def layer_forward(p, i, x, bias, num_heads, *, dropout_p=0.0, training=False):
    """One BertLayer. TF:models/bert/modeling_bert.py:282-351, 354-448."""
    pre = f"encoder.layer.{i}."
    H = x.size(-1)
    ctx = self_attention(p, pre, x, bias, num_heads, dropout_p=dropout_p, training=training)
    a = F.linear(ctx, p[pre + "attention.output.dense.weight"], p[pre + "attention.output.dense.bias"])
    a = _dropout(a, dropout_p, training)
    x1 = F.layer_norm(
        a + x, (H,), p[pre + "attention.output.LayerNorm.weight"], p[pre + "attention.output.LayerNorm.bias"], LN_EPS
    )
    f = F.linear(x1, p[pre + "intermediate.dense.weight"], p[pre + "intermediate.dense.bias"])
    f = F.gelu(f)  # exact erf GELU (hidden_act="gelu")
    o = F.linear(f, p[pre + "output.dense.weight"], p[pre + "output.dense.bias"])
    o = _dropout(o, dropout_p, training)
    return F.layer_norm(
        o + x1, (H,), p[pre + "output.LayerNorm.weight"], p[pre + "output.LayerNorm.bias"], LN_EPS
    )


def num_layers_of(p) -> int:
    return 1 + max(int(k.split(".")[2]) for k in p if k.startswith("encoder.layer."))


def encoder_forward(p, inputs_embeds, key_mask, num_heads, *, dropout_p=0.0, training=False, causal=True):
    """``BertModel(inputs_embeds=, attention_mask=).last_hidden_state``.

    TF:models/bert/modeling_bert.py:623-686. The pooler and the KV cache the
    reference also computes are unused by the training path and omitted.
    """
    x = embeddings_forward(p, inputs_embeds, dropout_p=dropout_p, training=training)
    bias = causal_padding_bias(key_mask, x.dtype, causal)
    for i in range(num_layers_of(p)):
        x = layer_forward(p, i, x, bias, num_heads, dropout_p=dropout_p, training=training)
    return x


def mean_pool(token_embeddings, key_mask):
    """sentence-transformers Pooling(mean): models.py:143-145."""
    m = key_mask.to(token_embeddings.dtype).unsqueeze(-1)
    return (token_embeddings * m).sum(1) / m.sum(1).clamp(min=1e-9)


def pool(token_embeddings, key_mask, mode: str = "mean"):
    """sentence-transformers ``Pooling(pooling_mode)`` for the modes ``ModelConfig.pooling_mode`` allows
    (``models.py:47, 143-145``). sentence-transformers 5.7.0 is not importable here: restated from its
    published behaviour -- mean: masked sum / clamp(mask sum, 1e-9); max: masked positions set to -1e9, then max;
    cls: token 0; lasttoken: the last position with mask 1 (position 0 when the row has none), times its mask.
    Parity of max / cls / lasttoken is therefore UNPINNED (no reference output exists for them in this container)."""
    if mode == "mean":
        return mean_pool(token_embeddings, key_mask)
    m = key_mask.to(torch.bool)
    if mode == "cls":
        return token_embeddings[:, 0]
    if mode == "max":
        return token_embeddings.masked_fill(~m[..., None], -1e9).max(1).values
    if mode == "lasttoken":
        L = key_mask.shape[1]
        values, indices = key_mask.long().flip(1).max(1)
        indices = torch.where(values == 0, torch.full_like(indices, L - 1), indices)
        gather = (L - indices - 1)[:, None, None].expand(-1, 1, token_embeddings.shape[-1])
        return torch.gather(token_embeddings * m[..., None].to(token_embeddings.dtype), 1, gather)[:, 0]
    raise ValueError(mode)


def algorithmic_flops_per_sequence(L, H, I, n_layers, n_neg_cols) -> float:
    """3*nL*(8LH^2 + 4LHI + 2L(L+1)H) + 4*L*M*H."""
    enc = 8 * L * H * H + 4 * L * H * I + 2 * L * (L + 1) * H
    return 3.0 * n_layers * enc + 4.0 * L * n_neg_cols * H


__all__ = [
    "LN_EPS",
    "init_params",
    "embeddings_forward",
    "causal_padding_bias",
    "self_attention",
    "layer_forward",
    "encoder_forward",
    "mean_pool",
    "algorithmic_flops_per_sequence",
    "math",
]
