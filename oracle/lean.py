"""Oracle at FULL batch size: the reference's loss sums evaluated per ITEM (TEST INFRASTRUCTURE).

The materialised oracle (``oracle.losses`` over the ``(Np, 1+N, H)`` candidates of ``oracle.model.compute_embeds``) is
the reference bit for bit but O(N^2 H): 21 GB at batch 32 already. At BASELINE config 2's benchmark batch (512 x 200 =
102 400 positions) even the ``(Np, 1+N)`` logits of the GEMM form are 4 x 10^10 numbers. Every per-column term of
every head is a function of the column's ITEM, and in-batch negatives repeat items (N = 102 400 draws from V = 3 883),
so

    sum_j f(l_ij)  =  sum_u mult_u f(l_iu)          (u over the catalogue, mult_u = how often item u was sampled)

turns the same sums into ``(Np, V+1)`` work that a CPU finishes in seconds. This file restates the heads of
``xfmr_rec/losses.py`` in that weighted form -- same formulas, ``weights = mult * negative_mask`` where the reference
has ``negative_mask`` -- and ``tests/test_oracle_golden.py::test_lean_by_item_equals_the_materialised_oracle`` pins it
against :mod:`oracle.losses` (itself pinned to the imported reference by the golden fixtures) on cases small enough
for both. Exact positive/negative ties are resolved by item id (``oracle.losses.embed_loss_parts(ties=...)``).

``rows_reference_form`` is the other half of the full-size check: for a SAMPLE of query rows it builds the reference's
``[rowdot(q, e_pos) | q E_neg^T]`` logits against the FULL negative list (no multiplicity identity involved) and runs
the unmodified heads of :mod:`oracle.losses` with autograd -- the rows' loss terms and ``dL/dq`` rows.
"""

from __future__ import annotations

import torch
import torch.nn.functional as F

from . import losses as L


def _wmean(values, weights):
    """losses.py:90-111 over dim 1 with real-valued weights."""
    return (values * weights / (weights.sum(dim=1, keepdim=True) + 1e-9)).sum(dim=1)


@torch.no_grad()
def heads_by_item(q, pos_items, neg_items, table, *, mask_false_negatives=True, scale=1.0, margin=0.5, chunk=4096):
    """All seven loss sums + LogitsStatistics of ``trainer.py:250-264`` for queries ``q`` (Np,H) whose positives are
    ``pos_items`` (Np) against the shared negative list ``neg_items`` (N item ids, repeats allowed).

    Follows losses.py:179-208 (dot / cosine logits), :263-293 (``logits < pos_logit``), :338-372, :408-543 (heads) and
    :375-405 (statistics) with ``weights = mult * mask``. Returns ``{"loss/<Class>": float, "logits/...": float}``."""
    V1 = table.shape[0]
    mult = torch.bincount(neg_items, minlength=V1).to(torch.float64)  # multiplicity of every catalogue row
    n_neg = int(neg_items.numel())
    tn = F.normalize(table, dim=-1, eps=1e-8)
    tot = {k: 0.0 for k in L.LOSS_KINDS}
    st = dict(dens=0.0, ps=0.0, pss=0.0, pmin=float("inf"), pmax=-float("inf"), ns=0.0, nss=0.0, nc=0.0,
              nmin=float("inf"), nmax=-float("inf"))
    for i0 in range(0, q.shape[0], chunk):
        qc, pc = q[i0:i0 + chunk], pos_items[i0:i0 + chunk]
        e_pos = table[pc]
        same = torch.arange(V1)[None, :] == pc[:, None]  # the column that IS the row's positive item
        for cosine in (False, True):
            if cosine:  # F.cosine_similarity (losses.py:206-208): normalised operands, eps 1e-8
                qn = F.normalize(qc, dim=-1, eps=1e-8)
                pos = (qn * tn[pc]).sum(-1, keepdim=True)
                lg = qn @ tn.T
            else:
                pos = (qc * e_pos).sum(-1, keepdim=True)
                lg = qc @ table.T
            lg = torch.where(same, pos, lg)  # ties by item id
            # losses.py:263-293 -- unmasked: every sampled column counts, also one that holds the positive's item
            mask = (lg < pos) if mask_false_negatives else torch.ones_like(same)
            w = mask.to(torch.float64) * mult[None, :]
            lg64, pos64 = lg.to(torch.float64), pos.to(torch.float64)
            if cosine:
                align = (1 - pos64[:, 0]).sum()
                contr = _wmean((lg64 - 1 + margin).relu(), w).sum()
                tot["AlignmentLoss"] += float(align)
                tot["ContrastiveLoss"] += float(contr)
                tot["AlignmentContrastiveLoss"] += float(align + contr)
                continue
            # InfoNCE (losses.py:479-488): CE over {positive} + counted negatives, logits scaled
            z = scale * lg64
            zmax = torch.maximum((z.masked_fill(w == 0, -float("inf"))).max(dim=1, keepdim=True).values, scale * pos64)
            lse = zmax[:, 0] + torch.log(torch.exp(scale * pos64[:, 0] - zmax[:, 0]) + (w * torch.exp(z - zmax)).sum(dim=1))
            tot["InfoNCELoss"] += float((lse - scale * pos64[:, 0]).sum())
            # NCE (losses.py:498-511): BCE with target 1 on the positive, 0 on the negatives
            tot["NCELoss"] += float((F.softplus(-pos64[:, 0]) + _wmean(F.softplus(lg64), w)).sum())
            sc = lg64 - pos64 * (1 - margin)  # losses.py:520-543
            tot["PairwiseHingeLoss"] += float(_wmean(sc.relu(), w).sum())
            tot["PairwiseLogisticLoss"] += float(_wmean(F.softplus(sc), w).sum())
            # LogitsStatistics (losses.py:375-405), dot logits
            cnt = w.sum(dim=1)
            st["dens"] += float((cnt / (n_neg + 1e-9)).sum())
            st["ps"] += float(pos64.sum()); st["pss"] += float((pos64 * pos64).sum())
            st["pmin"] = min(st["pmin"], float(pos64.min())); st["pmax"] = max(st["pmax"], float(pos64.max()))
            st["ns"] += float((w * lg64).sum()); st["nss"] += float((w * lg64 * lg64).sum()); st["nc"] += float(cnt.sum())
            sel = w > 0
            if bool(sel.any()):
                st["nmin"] = min(st["nmin"], float(lg64[sel].min())); st["nmax"] = max(st["nmax"], float(lg64[sel].max()))
    n = float(q.shape[0])
    out = {f"loss/{k}": v for k, v in tot.items()}
    out["logits/neg/density"] = st["dens"] / n
    out["logits/pos/mean"] = st["ps"] / n
    out["logits/pos/std"] = max(0.0, (st["pss"] - st["ps"] ** 2 / n) / (n - 1)) ** 0.5 if n > 1 else float("nan")
    out["logits/pos/min"], out["logits/pos/max"] = st["pmin"], st["pmax"]
    nc = st["nc"]
    out["logits/neg/count"] = nc
    if nc > 0:
        out["logits/neg/mean"] = st["ns"] / nc
        out["logits/neg/std"] = max(0.0, (st["nss"] - st["ns"] ** 2 / nc) / (nc - 1)) ** 0.5 if nc > 1 else float("nan")
        out["logits/neg/min"], out["logits/neg/max"] = st["nmin"], st["nmax"]
    return out


def rows_reference_form(kind, q_rows, pos_items, neg_items, table, *, mask_false_negatives=True, scale=1.0, margin=0.5):
    """The reference's logits ``[rowdot(q, e_pos) | q E_neg^T]`` (cosine heads: losses.py:206-208) of a SAMPLE
    of query rows against the full negative list, through the unmodified heads of :mod:`oracle.losses`. Returns
    ``(sum of the rows' loss terms, dL/dq_rows)``. Ties by item id, as ``oracle.model.lean_loss`` restores them."""
    q = q_rows.detach().clone().requires_grad_(True)
    e_pos, e_neg = table[pos_items], table[neg_items]
    if kind in L.COSINE_KINDS:
        qn, pn, nn_ = (F.normalize(t, dim=-1, eps=1e-8) for t in (q, e_pos, e_neg))
        logits = torch.cat([(qn * pn).sum(-1, keepdim=True), qn @ nn_.T], dim=1)
    else:
        logits = torch.cat([(q * e_pos).sum(-1, keepdim=True), q @ e_neg.T], dim=1)
    same = neg_items[None, :] == pos_items[:, None]
    logits = torch.cat([logits[:, :1], torch.where(same, logits[:, :1], logits[:, 1:])], dim=1)
    tgt = L.resolve_target(logits.size(0), None, "first", logits.device)
    mask = L.negative_mask(logits, tgt, mask_false_negatives)
    loss = L.head(kind, logits, tgt, mask, scale=scale, margin=margin)
    (g,) = torch.autograd.grad(loss, q)
    return loss.detach(), g
