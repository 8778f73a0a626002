#!/usr/bin/env python3
"""How far does "exact logit ties resolved by item id" (the kernels) move the loss and its gradient
from the reference's behaviour (ties kept or dropped by the rounding luck of its materialised bmm)? CPU experiment with
the oracle at a size the reference form can run: config 2's shape (V = 3883, L = 200, H = 128, 4 layers), batch 8, dense
rows, so that -- as in the benchmark -- a sizeable share of the sampled negatives of a row ARE its positive item.

    python scripts/tie_deviation.py [--batch 8]"""
import argparse
import pathlib
import sys

ROOT = pathlib.Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "transformer-recommenders_amd"):
    sys.path.insert(0, str(p))
import torch  # noqa: E402

from oracle import encoder as enc  # noqa: E402
from oracle import model as OM  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(8)
    B, L, H, V, nL, I = a.batch, 200, 128, 3883, 4, 512
    g = torch.Generator().manual_seed(0)
    table = torch.randn(V + 1, H, generator=g)
    table = table / table.norm(dim=-1, keepdim=True)
    table[0] = 0
    batch = {k: torch.randint(1, V + 1, (B, L), generator=g) for k in ("history_item_idx", "pos_item_idx", "neg_item_idx")}
    params0 = enc.init_params(H, nL, I, L, seed=0)
    res = {}
    for ties in (False, True):
        params = {k: v.clone().requires_grad_(True) for k, v in params0.items()}
        out = OM.compute_losses(params, table, batch, num_heads=H // 32, max_seq_length=L, loss_cfg={}, resolve_ties=ties)
        grads = {}
        for kind in ("InfoNCELoss", "PairwiseLogisticLoss", "AlignmentContrastiveLoss"):
            for p in params.values():
                p.grad = None
            out[f"loss/{kind}"].backward(retain_graph=True)
            grads[kind] = torch.cat([p.grad.flatten() for k, p in params.items() if not k.endswith("key.bias")])
        res[ties] = ({k: float(v) for k, v in out.items() if k.startswith("loss/") and not k.endswith("Mean")}, grads,
                     out["logits/neg/density"])
    N = B * L
    pos, neg = batch["pos_item_idx"].flatten(), batch["neg_item_idx"].flatten()
    tie_cols = int((neg[None, :] == pos[:, None]).sum())
    print(f"B={B}: {N} queries x {N} sampled negatives; {tie_cols} (query, negative) pairs are exact ties "
          f"({tie_cols / N:.2f} per query)")
    print(f"negative density (share of logits counted as negatives): reference luck {res[False][2]:.6f}, by item id {res[True][2]:.6f}")
    for k in res[False][0]:
        a_, b_ = res[False][0][k], res[True][0][k]
        print(f"  {k}: reference {a_:.6f}  by-id {b_:.6f}  rel diff {abs(a_ - b_) / max(abs(a_), 1e-12):.2e}")
    for kind, g0 in res[False][1].items():
        g1 = res[True][1][kind]
        print(f"  grad {kind}: rel-L2 diff {float((g0 - g1).norm() / g0.norm()):.2e}")


if __name__ == "__main__":
    main()
