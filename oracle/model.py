"""Oracle: model glue + training step (TEST INFRASTRUCTURE).

Restates the reference's ``RecommenderModel.forward`` / ``compute_embeds``
(``xfmr_rec/models.py:306-345, 366-419``) and the training-step driver
(``xfmr_rec/trainer.py:213-264, 288-291, 327-332``) over the functional
encoder in :mod:`oracle.encoder` and the loss pipeline in
:mod:`oracle.losses`.

``compute_embeds`` materialises the dense ``(Np, 1+N, H)`` candidate tensor
exactly as the reference does (``models.py:408-416``); this is what makes the
reference O(N^2 H) and is what the ``cpu_baseline`` leg of ``bench.py`` times.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

from . import encoder as enc
from . import losses as L


def build_table(item_embeddings: torch.Tensor) -> torch.Tensor:
    """Prepend the zero padding row. models.py:247-253."""
    return torch.cat([torch.zeros_like(item_embeddings[:1]), item_embeddings])


def forward(params, table, item_idx, *, num_heads, max_seq_length, dropout_p=0.0, training=False, causal=True):
    """models.py:306-345: truncate, gather, mask from embedding VALUES, encode, pool."""
    idx = item_idx[:, -max_seq_length:]
    x = F.embedding(idx, table)
    key_mask = (x != 0).any(-1).long()
    tok = enc.encoder_forward(params, x, key_mask, num_heads, dropout_p=dropout_p, training=training, causal=causal)
    return {
        "token_embeddings": tok,
        "sentence_embedding": enc.mean_pool(tok, key_mask),
        "attention_mask": key_mask,
    }


def compute_embeds(
    params, table, hist, pos, neg, *, num_heads, max_seq_length, is_normalized=False,
    dropout_p=0.0, training=False,
):
    """models.py:366-419."""
    out = forward(
        params, table, hist, num_heads=num_heads, max_seq_length=max_seq_length,
        dropout_p=dropout_p, training=training,
    )
    m = out["attention_mask"].bool()
    q = out["token_embeddings"][m]
    if is_normalized:
        q = F.normalize(q, dim=-1)
    pos_i = pos[m]
    pos_e = F.embedding(pos_i, table)[:, None, :]
    neg_e = F.embedding(neg[m], table)[None, :, :].expand(pos_e.size(0), -1, -1)
    cand = torch.cat([pos_e, neg_e], dim=1)
    keep = pos_i != 0
    # columns (>= 1) whose item IS the row's positive: exact ties, see oracle.losses.embed_loss_parts
    ties = torch.cat([torch.zeros_like(keep)[:, None], neg[m][None, :] == pos_i[:, None]], dim=1)
    return {
        "query_embed": q[keep],
        "candidate_embed": cand[keep],
        "attention_mask": m,
        "positive_mask": keep,
        "ties": ties[keep],
    }


def compute_losses(
    params, table, batch, *, num_heads, max_seq_length, loss_cfg, is_normalized=False,
    dropout_p=0.0, training=False, kinds=L.LOSS_KINDS, with_stats=True, resolve_ties=False,
):
    """trainer.py:213-264: all heads + batch statistics + logits statistics.

    ``resolve_ties``: resolve exact positive/negative ties by item identity (what the HIP kernel does) rather
    than by the rounding of the materialised bmm (what the reference does): oracle.losses.embed_loss_parts."""
    e = compute_embeds(
        params, table, batch["history_item_idx"], batch["pos_item_idx"], batch["neg_item_idx"],
        num_heads=num_heads, max_seq_length=max_seq_length, is_normalized=is_normalized,
        dropout_p=dropout_p, training=training,
    )
    am = e["attention_mask"]
    numel = am.numel()
    attn_nz = int(am.count_nonzero())
    pos_nz = int(e["positive_mask"].count_nonzero())
    out: dict = {}
    ties = e["ties"] if resolve_ties else None
    for kind in kinds:
        loss = L.embed_loss(kind, e["query_embed"], e["candidate_embed"], ties=ties, **loss_cfg)
        out[f"loss/{kind}"] = loss
        out[f"loss/{kind}Mean"] = loss / (pos_nz + 1e-9)
    out |= {
        "batch/size": am.size(0),
        "batch/seq_len": am.size(1),
        "batch/numel": numel,
        "batch/attention_non_zero": attn_nz,
        "batch/attention_density": attn_nz / (numel + 1e-9),
        "batch/positive_non_zero": pos_nz,
        "batch/positive_density": pos_nz / (attn_nz + 1e-9),
    }
    if with_stats:
        out |= L.logits_statistics(e["query_embed"], e["candidate_embed"], ties=ties, **loss_cfg)
    return out


class OracleTrainer:
    """Reference training loop on CPU: zero_grad -> training_step -> backward -> AdamW.step.

    trainer.py:288-291 (training_step), :327-332 (AdamW lr 1e-3, wd 0.01).
    """

    def __init__(
        self, params, table, *, num_heads, max_seq_length, train_loss="InfoNCELoss",
        loss_cfg=None, learning_rate=1e-3, weight_decay=0.01, dropout_p=0.0,
        faithful=True, is_normalized=False,
    ):
        self.params = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        self.table = table
        self.kw = dict(num_heads=num_heads, max_seq_length=max_seq_length, is_normalized=is_normalized)
        self.train_loss = train_loss
        self.loss_cfg = dict(loss_cfg or {})
        self.dropout_p = dropout_p
        self.faithful = faithful
        self.opt = torch.optim.AdamW(
            list(self.params.values()), lr=learning_rate, weight_decay=weight_decay
        )

    def training_step(self, batch):
        if not self.faithful and not self.kw.get("is_normalized"):
            # 'reference-lean' (BASELINE.md section 3): only the train head, GEMM-form logits, no statistics
            loss = lean_loss(self.params, self.table, batch, num_heads=self.kw["num_heads"],
                             max_seq_length=self.kw["max_seq_length"], kind=self.train_loss, loss_cfg=self.loss_cfg,
                             dropout_p=self.dropout_p)
            return loss, {f"loss/{self.train_loss}": loss}
        kinds = L.LOSS_KINDS if self.faithful else (self.train_loss,)
        out = compute_losses(
            self.params, self.table, batch, loss_cfg=self.loss_cfg, kinds=kinds,
            with_stats=self.faithful, dropout_p=self.dropout_p, training=self.dropout_p > 0,
            **self.kw,
        )
        return out[f"loss/{self.train_loss}"], out

    def step(self, batch):
        self.opt.zero_grad(set_to_none=True)
        loss, out = self.training_step(batch)
        loss.backward()
        self.opt.step()
        return loss.detach(), out


def lean_loss(params, table, batch, *, num_heads, max_seq_length, kind, loss_cfg, dropout_p=0.0):
    """'reference-lean' CPU variant: GEMM-form logits, one head.

    logits = [rowdot(q, e_pos) | Q E_neg^T]; mathematically identical to the
    materialised form, O(N^2) instead of O(N^2 H) memory.
    """
    out = forward(
        params, table, batch["history_item_idx"], num_heads=num_heads,
        max_seq_length=max_seq_length, dropout_p=dropout_p, training=dropout_p > 0,
    )
    m = out["attention_mask"].bool()
    q = out["token_embeddings"][m]
    pos_i = batch["pos_item_idx"][m]
    keep = pos_i != 0
    q = q[keep]
    e_pos = F.embedding(pos_i[keep], table)
    e_neg = F.embedding(batch["neg_item_idx"][m], table)
    if kind in L.COSINE_KINDS:
        qn, pn, nn_ = (F.normalize(t, dim=-1, eps=1e-8) for t in (q, e_pos, e_neg))
        logits = torch.cat([(qn * pn).sum(-1, keepdim=True), qn @ nn_.T], dim=1)
    else:
        logits = torch.cat([(q * e_pos).sum(-1, keepdim=True), q @ e_neg.T], dim=1)
    # A sampled negative that IS the row's positive item has, in the materialised form, bit-identical
    # logits in both columns (same vectors through the same bmm), so `logits < pos_logit` drops it.
    # rowdot and GEMM round differently, so the GEMM form restores the tie explicitly by item id.
    same_item = batch["neg_item_idx"][m][None, :] == pos_i[keep][:, None]
    logits[:, 1:] = torch.where(same_item, logits[:, :1], logits[:, 1:])
    tgt = L.resolve_target(logits.size(0), None, loss_cfg.get("target_position", "first"), logits.device)
    mask = L.negative_mask(logits, tgt, loss_cfg.get("mask_false_negatives", True))
    mask = L.hard_negative_mask(logits, mask, loss_cfg.get("num_hard_negatives", 0))
    return L.head(kind, logits, tgt, mask, scale=loss_cfg.get("scale", 1.0), margin=loss_cfg.get("margin", 0.5))
