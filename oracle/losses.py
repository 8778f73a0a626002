"""Oracle: embedding-loss pipeline and the seven loss heads (TEST INFRASTRUCTURE).

Functional restatement of ``xfmr_rec/losses.py`` (reference). The reference
organises this as an ``EmbedLoss`` class hierarchy; here every stage is a
free function over explicit arguments so the HIP kernels' intermediate
values (logits, masks, per-row terms) can be compared one by one.

``embed_loss(kind, q, cand, ...)`` is the whole ``EmbedLoss.forward``
(``losses.py:128-155``).
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

LOSS_KINDS = (
    "AlignmentLoss",
    "AlignmentContrastiveLoss",
    "ContrastiveLoss",
    "InfoNCELoss",
    "NCELoss",
    "PairwiseHingeLoss",
    "PairwiseLogisticLoss",
)
COSINE_KINDS = ("AlignmentLoss", "AlignmentContrastiveLoss", "ContrastiveLoss")


def dot_logits(q: torch.Tensor, cand: torch.Tensor) -> torch.Tensor:
    """(N,H) x (N,C,H) -> (N,C) row-wise dot products. losses.py:179-195."""
    return torch.bmm(q.unsqueeze(1), cand.transpose(1, 2)).squeeze(1)


def cosine_logits(q: torch.Tensor, cand: torch.Tensor) -> torch.Tensor:
    """Cosine similarity of each query with its candidates. losses.py:197-208."""
    return F.cosine_similarity(q.unsqueeze(1), cand, dim=-1)


def resolve_target(n_rows: int, target, target_position, device) -> torch.Tensor:
    """Column index of the positive per row, shape (N,1). losses.py:211-261."""
    if target is None and target_position is None:
        raise AssertionError("either `targets` or `config.target_position` must be provided")
    if target is not None and target_position is not None:
        raise AssertionError("only one of `targets` or `config.target_position` should be provided")
    if target_position == "first":
        target = torch.zeros(n_rows, dtype=torch.long, device=device)
    elif target_position == "diagonal":
        target = torch.arange(n_rows, dtype=torch.long, device=device)
    elif target_position is not None:
        raise ValueError(f"invalid target_position {target_position!r}")
    assert target.dim() == 1 and target.size(0) == n_rows
    return target.unsqueeze(1)


def negative_mask(logits, target, mask_false_negatives: bool) -> torch.Tensor:
    """True where a column counts as a negative. losses.py:263-293.

    With false-negative masking a column is a negative iff its logit is
    strictly below the positive's logit (this also drops the positive).
    Without it every column but the positive is a negative.
    """
    if not mask_false_negatives:
        m = torch.ones_like(logits, dtype=torch.bool)
        return m.scatter(1, target, False)
    return logits < logits.gather(1, target)


def hard_negative_mask(logits, neg_mask, num_hard_negatives: int) -> torch.Tensor:
    """Keep only the top-k negatives per row (k>0 and k<C). losses.py:295-330."""
    k = num_hard_negatives
    if k <= 0 or k >= logits.size(1):
        return neg_mask
    top = logits.masked_fill(~neg_mask, float("-inf")).topk(k, dim=1, sorted=False).indices
    keep = torch.zeros_like(neg_mask).scatter(1, top, True)
    return neg_mask & keep


def weighted_mean(values, weights, dim: int):
    """(v*w/(sum(w)+1e-9)).sum(dim). losses.py:90-111."""
    denom = weights.sum(dim=dim, keepdim=True) + 1e-9
    return (values * weights / denom).sum(dim=dim)


def _alignment(logits, target):  # losses.py:338-353
    return (1 - logits.gather(1, target)).sum()


def _contrastive(logits, neg_mask, margin):  # losses.py:355-372
    return weighted_mean((logits - 1 + margin).relu(), neg_mask, dim=1).sum()


def head(kind: str, logits, target, neg_mask, *, scale: float, margin: float):
    """The ``loss()`` of each head, given logits/target/mask. losses.py:408-543."""
    if kind == "AlignmentLoss":  # :420-426
        return _alignment(logits, target)
    if kind == "AlignmentContrastiveLoss":  # :442-447
        return _alignment(logits, target) + _contrastive(logits, neg_mask, margin)
    if kind == "ContrastiveLoss":  # :463-469
        return _contrastive(logits, neg_mask, margin)
    if kind == "InfoNCELoss":  # :479-488
        keep = neg_mask.scatter(1, target, True)
        z = logits.masked_fill(~keep, float("-inf")) * scale
        return F.cross_entropy(z, target[:, 0], reduction="sum")
    if kind == "NCELoss":  # :498-511
        y = torch.zeros_like(logits).scatter(1, target, 1.0)
        bce = F.binary_cross_entropy_with_logits(logits, y, reduction="none")
        pos = bce.gather(1, target)[:, 0]
        return (pos + weighted_mean(bce, neg_mask, dim=1)).sum()
    if kind in ("PairwiseHingeLoss", "PairwiseLogisticLoss"):  # :520-543
        scores = logits - logits.gather(1, target) * (1 - margin)
        act = scores.relu() if kind == "PairwiseHingeLoss" else F.softplus(scores)
        return weighted_mean(act, neg_mask, dim=1).sum()
    raise ValueError(kind)


def embed_loss_parts(
    kind: str,
    q: torch.Tensor,
    cand: torch.Tensor,
    target: torch.Tensor | None = None,
    *,
    target_position="first",
    mask_false_negatives: bool = True,
    num_hard_negatives: int = 0,
    scale: float = 1.0,
    margin: float = 0.5,
    ties: torch.Tensor | None = None,
):
    """``EmbedLoss.forward`` (losses.py:128-155) returning every intermediate.

    ``ties`` (optional bool (N,C)): columns that hold the SAME item vector as the row's target. Their logit
    equals the target's in exact arithmetic, so ``logits < target_logit`` must drop them; in the reference
    whether that happens depends on the rounding of its ``bmm`` (column 0 and column j are reduced in
    different vector lanes: observed both ways, see tests/golden/g4 'ties'). Passing ``ties`` makes the
    outcome independent of that luck by copying the target's logit into those columns -- the semantics the
    HIP kernel implements by item id. Without ``ties`` this function is the reference, bit for bit.
    """
    assert q.dim() == 2 and cand.dim() == 3  # losses.py:157-177
    assert q.size(0) == cand.size(0) and q.size(-1) == cand.size(-1)
    logits = cosine_logits(q, cand) if kind in COSINE_KINDS else dot_logits(q, cand)
    tgt = resolve_target(logits.size(0), target, target_position, logits.device)
    if ties is not None:
        logits = torch.where(ties, logits.gather(1, tgt), logits)
    mask = negative_mask(logits, tgt, mask_false_negatives)
    mask = hard_negative_mask(logits, mask, num_hard_negatives)
    loss = head(kind, logits, tgt, mask, scale=scale, margin=margin)
    return {"loss": loss, "logits": logits, "target": tgt, "negative_mask": mask}


def embed_loss(kind, q, cand, target=None, **cfg) -> torch.Tensor:
    return embed_loss_parts(kind, q, cand, target, **cfg)["loss"]


def logits_statistics(
    q,
    cand,
    target=None,
    *,
    target_position="first",
    mask_false_negatives: bool = True,
    num_hard_negatives: int = 0,
    ties: torch.Tensor | None = None,
    **_unused,
) -> dict[str, float]:
    """``LogitsStatistics`` (losses.py:375-405): dot logits, monitoring only. ``ties``: see embed_loss_parts."""
    logits = dot_logits(q, cand)
    tgt = resolve_target(logits.size(0), target, target_position, logits.device)
    if ties is not None:
        logits = torch.where(ties, logits.gather(1, tgt), logits)
    mask = hard_negative_mask(
        logits, negative_mask(logits, tgt, mask_false_negatives), num_hard_negatives
    )
    num_neg = mask.size(1) - 1
    if num_hard_negatives > 0:
        num_neg = min(num_neg, num_hard_negatives)
    stats = {"logits/neg/density": (mask.sum(dim=1) / (num_neg + 1e-9)).mean().item()}
    for key, v in (("pos", logits.gather(1, tgt)), ("neg", logits[mask])):
        if v.numel() > 0:
            stats[f"logits/{key}/mean"] = v.mean().item()
            stats[f"logits/{key}/std"] = v.std().item()
            stats[f"logits/{key}/min"] = v.min().item()
            stats[f"logits/{key}/max"] = v.max().item()
    return stats
