"""Exact GPU retrieval + ranking metrics for the validation path.

Mirrors what ``RecommenderLightningModule.recommend / predict_step / compute_metrics`` do through LanceDB and
torchmetrics (``xfmr_rec/trainer.py:186-211, 266-314``, ``index.py:214-255``, ``metrics.py:17-79``), for a batch of
users at once: ``ExactItemIndex.search`` = exact top-k over the item table with each user's history excluded;
``compute_retrieval_metrics`` = the seven metrics under the reference's names.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _native as N

METRICS = {"cosine": 0, "dot": 1, "l2": 2}
METRIC_NAMES = (  # metrics.py:7-15, in the order of XFMR_RM_*
    "retrieval_normalized_dcg", "retrieval_average_precision", "retrieval_auroc", "retrieval_precision",
    "retrieval_recall", "retrieval_hit_rate", "retrieval_reciprocal_rank",
)


def _csr(lists, device):
    lens = np.asarray([len(x) for x in lists], dtype=np.int64)
    off = np.zeros(len(lists) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    flat = np.concatenate([np.asarray(x, dtype=np.int64) for x in lists]) if off[-1] else np.zeros(1, dtype=np.int64)
    return torch.from_numpy(flat).to(device), torch.from_numpy(off).to(device)


class ExactItemIndex:
    """``LanceIndex.search`` (``index.py:214-255``) as an exact scan of the item table (row 0 = padding)."""

    def __init__(self, table: torch.Tensor, table_rnorm: torch.Tensor | None = None, index_metric: str = "cosine"):
        from . import ops

        self.table = table.contiguous()
        self.rnorm = table_rnorm if table_rnorm is not None else ops.table_rnorm(self.table)
        self.metric = METRICS[index_metric]

    def search(self, embedding: torch.Tensor, exclude_item_idx=None, top_k: int = 20):
        """embedding (B,H) or (H,); exclude_item_idx: per-query lists of item indices (the users' histories).
        Returns (item_idx (B,top_k) int64, -1 padded; score (B,top_k) = 1 - distance), best first."""
        q = embedding.reshape(-1, embedding.shape[-1]).contiguous().to(torch.float32)
        B, H = q.shape
        dev = q.device
        idx = torch.empty((B, top_k), dtype=torch.int64, device=dev)
        score = torch.empty((B, top_k), dtype=torch.float32, device=dev)
        ex, exo = (None, None) if exclude_item_idx is None else _csr(exclude_item_idx, dev)
        lib = N.load()
        nbytes = lib.xfmr_topk_workspace(B, self.table.shape[0])
        ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        N.check(
            lib.xfmr_topk(N.ptr(q), N.ptr(self.table), N.ptr(self.rnorm), self.table.shape[0], B, H, N.ptr(ex),
                          N.ptr(exo), top_k, self.metric, N.ptr(idx), N.ptr(score), N.ptr(ws), nbytes, N.stream()),
            "xfmr_topk",
        )
        return idx, score


def retrieval_metrics(rec_idx: torch.Tensor, target_idx, top_k: int):
    """(values (B,7) on the device in the order of METRIC_NAMES, valid (B,) bool)."""
    B, k = rec_idx.shape
    dev = rec_idx.device
    tg, tgo = _csr(target_idx, dev)
    out = torch.empty((B, len(METRIC_NAMES)), dtype=torch.float32, device=dev)
    valid = torch.empty((B,), dtype=torch.uint8, device=dev)
    N.check(
        N.load().xfmr_retrieval_metrics(N.ptr(rec_idx.contiguous()), N.ptr(tg), N.ptr(tgo), B, k, top_k, N.ptr(out),
                                        N.ptr(valid), N.stream()),
        "xfmr_retrieval_metrics",
    )
    return out, valid.bool()


def compute_retrieval_metrics(rec_idx, target_idx, top_k: int) -> dict[str, torch.Tensor]:
    """``metrics.py:17-79`` for ONE ranked list (item indices instead of id strings): {} when there is no target."""
    rec = torch.as_tensor(rec_idx, dtype=torch.int64, device="cuda").reshape(1, -1)
    vals, valid = retrieval_metrics(rec, [list(target_idx)], top_k)
    if not bool(valid[0]):
        return {}
    return {name: vals[0, i] for i, name in enumerate(METRIC_NAMES)}
