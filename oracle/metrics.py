"""Oracle: exact top-k and the seven retrieval metrics of the validation path (TEST INFRASTRUCTURE).

``compute_retrieval_metrics`` restates ``xfmr_rec/metrics.py:17-79`` with the definitions of
``torchmetrics.functional.retrieval`` 1.9.0 (``uv.lock``) written out per list; torchmetrics (and lancedb) are not
importable in this container, so the metric VALUES are **parity unpinned**: they follow the published
definitions (nDCG with linear gains and 1/log2(rank+1) discounts; AP as the mean precision at the hits inside the
cutoff; AUROC over the cutoff's positives/negatives, 0 when one class is missing; precision / top_k; recall / number of
targets; hit rate; reciprocal rank of the first hit inside the cutoff).
"""

from __future__ import annotations

import math

import numpy as np


def topk(query, table, exclude, k, metric="cosine"):
    """Exact search of ``index.py:214-255``: best-first item indices (row 0 = padding, never returned), -1 padded."""
    q = np.asarray(query, dtype=np.float64)
    t = np.asarray(table, dtype=np.float64)
    dot = t @ q
    if metric == "cosine":
        s = dot / (max(np.linalg.norm(q), 1e-8) * np.maximum(np.linalg.norm(t, axis=1), 1e-8))
    elif metric == "dot":
        s = dot
    else:
        s = 1.0 - ((t - q[None]) ** 2).sum(1)
    s = s.copy()
    s[0] = -np.inf
    for x in exclude or ():
        s[int(x)] = -np.inf
    order = np.lexsort((np.arange(len(s)), -s))  # score descending, then index ascending
    order = [int(i) for i in order if np.isfinite(s[i])][:k]
    return order + [-1] * (k - len(order)), [float(s[i]) for i in order]


def compute_retrieval_metrics(rec_ids, target_ids, top_k):
    """metrics.py:17-79 on one ranked list of ids (any hashable; ``-1`` / ``""`` = padding)."""
    if len(target_ids) == 0:
        return {}
    rec = list(rec_ids)
    if len(rec) < top_k:
        rec = rec + [None] * (top_k - len(rec))
    targets = set(target_ids)
    all_items = rec + [t for t in targets if t not in set(rec_ids)]
    target = [item in targets for item in all_items]  # preds = linspace(1, 0): the list order IS the ranking
    k = min(top_k, len(all_items))
    top = target[:k]
    hits = sum(top)
    n_pos = sum(target)
    dcg = sum(1.0 / math.log2(i + 2) for i, r in enumerate(top) if r)
    idcg = sum(1.0 / math.log2(i + 2) for i in range(min(n_pos, k)))
    pos_ranks = [i + 1 for i, r in enumerate(top) if r]
    ap = float(np.mean([(j + 1) / p for j, p in enumerate(pos_ranks)])) if pos_ranks else 0.0
    n_neg = k - hits
    auroc = 0.0
    if hits > 0 and n_neg > 0:
        above = 0
        pairs = 0
        for r in top:
            if r:
                above += 1
            else:
                pairs += above
        auroc = pairs / (hits * n_neg)
    return {
        "retrieval_normalized_dcg": dcg / idcg if idcg > 0 else 0.0,
        "retrieval_average_precision": ap,
        "retrieval_auroc": auroc,
        "retrieval_precision": hits / top_k,
        "retrieval_recall": hits / n_pos,
        "retrieval_hit_rate": 1.0 if hits > 0 else 0.0,
        "retrieval_reciprocal_rank": 1.0 / pos_ranks[0] if pos_ranks else 0.0,
    }
