"""Validation path: exact top-k with history exclusion and the seven retrieval metrics.

CPU: the oracle's metric restatement on hand-computed cases (the reference's docstring example included).
GPU: xfmr_topk against numpy argsort (all three metrics, exclusions, ties, k larger than the catalogue) and
xfmr_retrieval_metrics against the oracle on random ranked lists.
"""

import numpy as np
import pytest
import torch

from oracle import metrics as OMx


def test_oracle_metrics_on_hand_computed_cases():
    m = OMx.compute_retrieval_metrics(["i10", "i3", "i7"], {"i3"}, top_k=3)  # metrics.py docstring example
    assert m["retrieval_auroc"] == pytest.approx(0.5)  # one negative above the hit, one below
    assert m["retrieval_reciprocal_rank"] == pytest.approx(0.5)
    assert m["retrieval_normalized_dcg"] == pytest.approx((1 / np.log2(3)) / 1.0)
    assert m["retrieval_precision"] == pytest.approx(1 / 3) and m["retrieval_recall"] == 1.0
    assert m["retrieval_average_precision"] == pytest.approx(0.5) and m["retrieval_hit_rate"] == 1.0
    assert OMx.compute_retrieval_metrics(["a"], [], top_k=3) == {}
    m = OMx.compute_retrieval_metrics(["a", "b"], {"z", "b"}, top_k=4)  # padded list, one target never retrieved
    assert m["retrieval_recall"] == pytest.approx(0.5) and m["retrieval_precision"] == pytest.approx(0.25)
    assert m["retrieval_normalized_dcg"] == pytest.approx((1 / np.log2(3)) / (1 + 1 / np.log2(3)))
    m = OMx.compute_retrieval_metrics(["a", "b"], {"q"}, top_k=2)
    assert all(v == 0.0 for v in m.values())


@pytest.mark.gpu
@pytest.mark.parametrize("metric,H", [("cosine", 64), ("dot", 64), ("l2", 64), ("cosine", 384), ("l2", 384)])
def test_topk_matches_exact_numpy_search(metric, H):
    from xfmr_rec_amd.retrieval import ExactItemIndex

    g = torch.Generator().manual_seed(0)
    V, B, k = 777, 9, 20
    table = torch.randn(V + 1, H, generator=g)
    table[0] = 0
    table[5] = table[9]  # exact ties
    q = torch.randn(B, H, generator=g)
    excl = [sorted(set(torch.randint(1, V + 1, (int(n),), generator=g).tolist())) for n in [0, 1, 5, 50, 300, 3, 3, 3, 776]]
    excl[8] = list(range(1, V + 1))[:770]  # fewer than k items remain -> -1 padding
    idx, score = ExactItemIndex(table.cuda(), index_metric=metric).search(q.cuda(), excl, top_k=k)
    idx, score = idx.cpu().numpy(), score.cpu().numpy()
    for b in range(B):
        want_idx, want_s = OMx.topk(q[b].numpy(), table.numpy(), excl[b], k, metric)
        n = len(want_s)
        np.testing.assert_allclose(score[b, :n], want_s, rtol=2e-5, atol=2e-5)
        assert (idx[b, n:] == -1).all()
        assert not (set(idx[b, :n].tolist()) & set(excl[b])) and 0 not in idx[b, :n]
        # same set up to numerically tied scores at the boundary
        diff = set(idx[b, :n].tolist()) ^ set(want_idx[:n])
        assert len(diff) <= 2, (b, diff)
        assert (np.diff(score[b, :n]) <= 1e-7).all()


@pytest.mark.gpu
def test_retrieval_metrics_match_the_oracle():
    from xfmr_rec_amd import retrieval as R

    rng = np.random.default_rng(0)
    B, k, top_k = 64, 20, 20
    rec = np.stack([rng.permutation(200)[:k] + 1 for _ in range(B)]).astype(np.int64)
    rec[3, 12:] = -1  # a short list
    targets = [list(rng.integers(1, 201, int(n))) for n in rng.integers(0, 9, B)]
    targets[5] = list(rec[5, :3]) + list(rec[5, :2])  # duplicates, all at the top
    targets[6] = []
    vals, valid = R.retrieval_metrics(torch.from_numpy(rec).cuda(), targets, top_k)
    vals, valid = vals.cpu().numpy(), valid.cpu().numpy()
    for b in range(B):
        want = OMx.compute_retrieval_metrics([int(x) for x in rec[b] if x >= 0], [int(t) for t in targets[b]], top_k)
        assert bool(valid[b]) == bool(want)
        for i, name in enumerate(R.METRIC_NAMES):
            if want:
                assert vals[b, i] == pytest.approx(want[name], rel=1e-5, abs=1e-6), (b, name)
    one = R.compute_retrieval_metrics(rec[0], targets[0] or [int(rec[0, 1])], top_k)
    assert set(one) == set(R.METRIC_NAMES)


@pytest.mark.gpu
def test_validation_step_mirrors_the_reference_surface():
    """recommend / predict_step / compute_metrics / validation_step (trainer.py:186-314) on a tiny model: the
    history is never recommended, the keys are the reference's, and the values equal the oracle's metrics on the
    oracle's exact search of the same pooled embedding."""
    import xfmr_rec_amd as X

    V, H = 60, 64
    g = torch.Generator().manual_seed(2)
    table = torch.randn(V + 1, H, generator=g)
    table[0] = 0
    conf = X.LightningConfig(hidden_size=H, num_attention_heads=2, intermediate_size=64, num_hidden_layers=1,
                             max_seq_length=16, precision="fp32", top_k=10)
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.set_table(table.cuda())
    mod = mod.to("cuda").eval()
    hist = [3, 7, 9, 21, 40]
    row = {"history": {"item_id": np.array(hist)},
           "target": {"item_id": np.array([5, 8, 13, 50]), "label": np.array([True, False, True, True])}}
    rec = mod.predict_step(row)
    got_idx = rec["item_idx"].cpu().numpy()
    assert not (set(got_idx.tolist()) & set(hist)) and 0 not in got_idx and len(set(got_idx.tolist())) == 10
    emb = mod.model(torch.tensor([hist]).cuda())["sentence_embedding"][0].cpu().numpy()
    want_idx, _ = OMx.topk(emb, table.numpy(), hist, 10, "cosine")
    assert got_idx.tolist() == want_idx
    out = mod.validation_step(row)
    want = OMx.compute_retrieval_metrics(want_idx, [5, 13, 50], 10)
    assert set(out) == {f"val/{k}" for k in want}
    for k, v in want.items():
        assert float(out[f"val/{k}"]) == pytest.approx(v, rel=1e-5, abs=1e-6)
    assert "val/retrieval_normalized_dcg" in out  # the monitored metric (params.py:12)
