"""Embedding losses: host-side mirror of the reference's ``xfmr_rec/losses.py`` over the fused HIP kernel.

Same class names, config fields, ``forward(query_embed, candidate_embed, target=None)`` signature and
assertion behaviour as the reference (``losses.py:11-30, 114-155, 408-564``). The arithmetic runs in
``libxfmr_hip.so`` (``xfmr_sampled_loss_lists``): logits are never materialised.

Candidates. The reference's training path always passes the ``(Np, 1+N, H)`` tensor that
``RecommenderModel.compute_embeds`` builds by ``expand`` + ``cat`` (``models.py:398-416``): column 0 is the
row's positive item, columns 1..N are the same N sampled negatives for every row. That tensor is O(N^2 H)
bytes (21 GB at batch 32 x 200 x 128), so here ``compute_embeds`` returns it *structured* instead of
materialised:

* :class:`SharedNegatives` -- ``[E[pos_items] | E[neg_items]]`` (positive at column 0, i.e.
  ``target_position="first"``),
* :class:`CatalogCandidates` -- every row of the item table as a column, the row's positive given by
  ``target`` (``target_position=None``): full-catalogue softmax.

Both report the shape / dim of the dense tensor they stand for and can ``materialize()`` it.
"""

from __future__ import annotations

import abc
from typing import Literal

import pydantic
import torch

from . import _native as N
from . import ops


class LossConfig(pydantic.BaseModel):
    """Same fields and defaults as the reference (``losses.py:11-30``)."""

    target_position: Literal["first", "diagonal"] | None = "first"
    mask_false_negatives: bool = True
    num_hard_negatives: int = 0
    scale: float = 1.0
    margin: float = 0.5


class _StructuredCandidates:
    """Stands for the reference's dense ``(Np, C, H)`` candidate tensor (``models.py:408-416``) without holding it. What a
    caller of the reference's ``compute_embeds`` does with that tensor works here too: ``shape`` / ``size()`` / ``dim()`` /
    ``dtype`` / ``device``, indexing -- ``cand[i]``, ``cand[a:b]``, ``cand[rows, cols]`` build only the rows asked for --
    and ``materialize()`` / ``torch.as_tensor(cand.materialize())`` for the whole thing (O(Np * C * H): small inputs)."""

    table: torch.Tensor
    table_rnorm: torch.Tensor

    def dim(self) -> int:
        return 3

    def size(self, i: int | None = None):
        return self.shape if i is None else self.shape[i]

    @property
    def device(self):
        return self.table.device

    @property
    def dtype(self):
        return self.table.dtype

    def __len__(self) -> int:
        return int(self.shape[0])

    def _rows(self, rows: torch.Tensor) -> torch.Tensor:  # (len(rows), C, H) dense
        raise NotImplementedError

    def __getitem__(self, idx):
        first, rest = (idx[0], idx[1:]) if isinstance(idx, tuple) else (idx, ())
        n = int(self.shape[0])
        if isinstance(first, int):
            out = self._rows(torch.as_tensor([first % n], device=self.device))[0]
        elif first is Ellipsis:
            return self.materialize()[idx]
        else:
            out = self._rows(torch.arange(n, device=self.device)[first])
            rest = (slice(None),) + tuple(rest)
        return out[tuple(rest)] if rest else out


class SharedNegatives(_StructuredCandidates):
    """``cat([E[pos_items][:, None], E[neg_items][None].expand(Np, -1, -1)], 1)`` without the copy."""

    def __init__(self, table, table_rnorm, pos_items, neg_items, table_bf16=None):
        self.table, self.table_rnorm, self.table_bf16 = table, table_rnorm, table_bf16
        self.pos_items, self.neg_items = pos_items.contiguous(), neg_items.contiguous()

    @property
    def shape(self):
        return torch.Size((self.pos_items.numel(), 1 + self.neg_items.numel(), self.table.shape[1]))

    def _rows(self, rows):
        pos = self.table[self.pos_items[rows]][:, None, :]
        neg = self.table[self.neg_items][None, :, :].expand(pos.size(0), -1, -1)
        return torch.cat([pos, neg], dim=1)

    def materialize(self) -> torch.Tensor:
        """The dense tensor of ``models.py:408-416`` (debug / small inputs only: O(Np*N*H))."""
        return self._rows(torch.arange(self.pos_items.numel(), device=self.pos_items.device))


class CatalogCandidates(_StructuredCandidates):
    """``table[None].expand(Np, -1, -1)``: every item is a column; ``target`` names the positive."""

    def __init__(self, table, table_rnorm, n_query: int, table_bf16=None):
        self.table, self.table_rnorm, self.n_query, self.table_bf16 = table, table_rnorm, int(n_query), table_bf16

    @property
    def shape(self):
        return torch.Size((self.n_query, self.table.shape[0], self.table.shape[1]))

    def _rows(self, rows):
        return self.table[None].expand(int(rows.numel()), -1, -1)

    def materialize(self) -> torch.Tensor:
        return self.table[None].expand(self.n_query, -1, -1)


class EmbedLoss(torch.nn.Module, abc.ABC):
    """Base class: shape checks, target handling, dispatch to the fused kernel (``losses.py:114-155``)."""

    def __init__(self, config: LossConfig, *, precision: str = "bf16") -> None:
        super().__init__()
        self.config = config
        self.precision = getattr(config, "precision", precision)

    @property
    def kind(self) -> str:
        return self.__class__.__name__

    # -- reference pipeline stages that still make sense without a logits tensor ---------------------
    def check_embeds(self, query_embed, candidate_embed) -> None:
        """Same assertions as ``losses.py:157-177``."""
        assert query_embed.dim() == 2, f"{query_embed.dim() = }, {query_embed.size() = }"
        assert candidate_embed.dim() == 3, f"{candidate_embed.dim() = }, {candidate_embed.size() = }"
        assert query_embed.size(0) == candidate_embed.size(0), (
            f"{query_embed.size(0) = } != {candidate_embed.size(0) = }"
        )
        assert query_embed.size(-1) == candidate_embed.size(-1), (
            f"{query_embed.size(-1) = } != {candidate_embed.size(-1) = }"
        )

    def check_target(self, n_rows: int, target):
        """Same contract as ``losses.py:211-261`` (exactly one of target / target_position)."""
        assert target is not None or self.config.target_position is not None, (
            "either `targets` or `config.target_position` must be provided"
        )
        assert target is None or self.config.target_position is None, (
            "only one of `targets` or `config.target_position` should be provided"
        )
        if self.config.target_position not in (None, "first", "diagonal"):
            msg = f"invalid {self.config.target_position = }"
            raise ValueError(msg)
        if target is not None:
            assert target.dim() == 1, f"{target.dim() = }, {target.size() = }"
            assert target.size(0) == n_rows, f"{target.size(0) = } != {n_rows = }"
        return target

    def _opts(self, mode: int, *, all_heads: bool = False) -> dict:
        c = self.config
        return dict(
            train_head=self.kind if self.kind in N.LOSS_IDS else "InfoNCELoss", all_heads=all_heads,
            mask_false_negatives=c.mask_false_negatives, mode=mode, scale=c.scale, margin=c.margin,
            precision=self.precision, num_hard_negatives=c.num_hard_negatives,
        )

    def _run(self, query_embed, candidate_embed, target, *, all_heads: bool = False):
        self.check_embeds(query_embed, candidate_embed)
        target = self.check_target(query_embed.size(0), target)
        q = query_embed.contiguous().to(torch.float32)
        if isinstance(candidate_embed, SharedNegatives):
            if self.config.target_position != "first":
                raise ValueError("SharedNegatives candidates put the positive at column 0: target_position='first'")
            return ops.sampled_loss_lists_train(
                q, candidate_embed.pos_items, candidate_embed.neg_items, candidate_embed.table,
                candidate_embed.table_rnorm,
                self._opts(N.NEG_SHARED, all_heads=all_heads) | {"table_bf16": candidate_embed.table_bf16},
            )
        if isinstance(candidate_embed, CatalogCandidates):
            if target is None:
                raise ValueError("CatalogCandidates need `target` (the positive item index per row)")
            opts = self._opts(N.NEG_CATALOG, all_heads=all_heads) | {"table_bf16": candidate_embed.table_bf16}
            if opts["mask_false_negatives"]:
                # the reference would additionally mask catalogue items scoring above the positive; supported
                pass
            return ops.sampled_loss_lists_train(
                q, target.contiguous(), None, candidate_embed.table, candidate_embed.table_rnorm, opts
            )
        # the reference's own calling convention: a dense (N,C,H) tensor (xfmr_dense_loss, C <= 8192). The training
        # path never builds it -- compute_embeds returns the structured forms above.
        c = self.config
        opts = dict(
            target_position=c.target_position, train_head=self.kind if self.kind in N.LOSS_IDS else "InfoNCELoss",
            all_heads=all_heads, mask_false_negatives=c.mask_false_negatives,
            num_hard_negatives=c.num_hard_negatives, scale=c.scale, margin=c.margin,
        )
        tgt = None if target is None else target.contiguous().to(torch.int64)
        return ops.dense_loss_train(q, candidate_embed.contiguous().to(torch.float32), tgt, opts)

    def forward(self, query_embed, candidate_embed, target=None):
        """Summed loss over the batch (``losses.py:128-155``)."""
        loss, _all, _stats = self._run(query_embed, candidate_embed, target)
        return loss


class LogitsStatistics(EmbedLoss):
    """Monitoring statistics of the dot-product logits (``losses.py:375-405``): returns floats."""

    def forward(self, query_embed, candidate_embed, target=None) -> dict[str, float]:
        with torch.no_grad():
            _loss, _all, stats = self._run(query_embed, candidate_embed, target, all_heads=True)
        return stats_to_dict(stats.tolist())


def stats_to_dict(s: list[float]) -> dict[str, float]:
    """Device statistics vector -> the reference's key names (``losses.py:392-404``)."""
    out = {"logits/neg/density": s[N.STAT["neg_density"]]}
    if s[N.STAT["n_query"]] > 0:
        out |= {f"logits/pos/{k}": s[N.STAT[f"pos_{k}"]] for k in ("mean", "std", "min", "max")}
    if s[N.STAT["neg_count"]] > 0:
        out |= {f"logits/neg/{k}": s[N.STAT[f"neg_{k}"]] for k in ("mean", "std", "min", "max")}
    return out


class AlignmentLoss(EmbedLoss):
    """sum_i (1 - cos(q_i, pos_i))  (``losses.py:408-426``)."""


class AlignmentContrastiveLoss(EmbedLoss):
    """alignment + margin contrastive on cosine logits: the 'CCL' head (``losses.py:429-447``)."""


class ContrastiveLoss(EmbedLoss):
    """sum_i mean_{j in neg_i} relu(cos_ij - 1 + margin)  (``losses.py:450-469``)."""


class InfoNCELoss(EmbedLoss):
    """masked, scaled cross-entropy: sampled softmax / SSM (``losses.py:472-488``)."""


class NCELoss(EmbedLoss):
    """binary NCE with sigmoid cross-entropy (``losses.py:491-511``)."""


class PairwiseHingeLoss(EmbedLoss):
    """sum_i mean_j relu(l_ij - l_ip (1 - margin))  (``losses.py:514-527``)."""


class PairwiseLogisticLoss(EmbedLoss):
    """sum_i mean_j softplus(l_ij - l_ip (1 - margin)): BPR at margin 0 (``losses.py:530-543``)."""


LOSS_CLASSES: list[type[EmbedLoss]] = [
    AlignmentLoss,
    AlignmentContrastiveLoss,
    ContrastiveLoss,
    InfoNCELoss,
    NCELoss,
    PairwiseHingeLoss,
    PairwiseLogisticLoss,
]

LossType = Literal[
    "AlignmentLoss",
    "AlignmentContrastiveLoss",
    "ContrastiveLoss",
    "InfoNCELoss",
    "NCELoss",
    "PairwiseHingeLoss",
    "PairwiseLogisticLoss",
]
