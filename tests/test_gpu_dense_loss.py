"""GPU parity of the dense-candidate EmbedLoss form (xfmr_dense_loss) against every case of golden set G1 --
produced by the reference's own ``xfmr_rec/losses.py`` (oracle/make_golden.py): 7 heads x {default,
mask_false_negatives=False, num_hard_negatives, scale=20, margin=0, target_position diagonal / explicit target}
-- and against the CPU oracle on larger seeded inputs (duplicated candidates, i.e. ties at the top-k threshold).

Called through the reference-shaped class API (``xfmr_rec_amd.losses.<Head>(LossConfig)(query, candidates,
target)``), which goes through the C ABI. fp32 vector arithmetic: loss rel <= 1e-5, dq <= 1e-5 * max(1,|x|).
"""

import json

import numpy as np
import pytest
import torch

from oracle import losses as OL

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a))


def _grad_close(got, want, what):
    scale = max(float(np.abs(want).max()), 1e-6)
    assert float(np.linalg.norm(got - want)) <= 1e-4 * max(float(np.linalg.norm(want)), 1e-6), what
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-4 * scale, err_msg=what)


@pytest.fixture(scope="module")
def L():
    import xfmr_rec_amd.losses as m

    return m


@pytest.fixture(scope="module")
def g1(golden_dir):
    return np.load(golden_dir / "g1_losses.npz")


def test_dense_loss_matches_reference_goldens(L, g1):
    index = json.loads(str(g1["index"]))
    seen = set()
    for c in index:
        n = c["shape"][0]
        q = _t(g1[f"in/s{c['seed']}_n{n}/q"]).to(DEV).requires_grad_(True)
        cand = _t(g1[f"in/s{c['seed']}_n{n}/cand"]).to(DEV)
        target = _t(g1[f"in/s{c['seed']}_n{n}/target"]).to(DEV) if c["has_target"] else None
        head = getattr(L, c["kind"])(L.LossConfig(**c["cfg"]))
        loss = head(q, cand, target)
        want = float(g1[f"{c['key']}/loss"])
        assert loss.item() == pytest.approx(want, rel=1e-5, abs=1e-5), (c["key"], c["kind"], c["cfg"])
        loss.backward()
        dq = g1[f"{c['key']}/dq"]
        _grad_close(q.grad.cpu().numpy(), dq, str(c))
        seen.add(c["kind"])
    assert seen == set(OL.LOSS_KINDS)


def test_dense_logits_statistics_match_reference_goldens(L, g1):
    index = json.loads(str(g1["index"]))
    done = set()
    for c in index:
        n = c["shape"][0]
        key = f"stats/s{c['seed']}_n{n}_v{c['variant']}"
        if key in done:
            continue
        done.add(key)
        q = _t(g1[f"in/s{c['seed']}_n{n}/q"]).to(DEV)
        cand = _t(g1[f"in/s{c['seed']}_n{n}/cand"]).to(DEV)
        target = _t(g1[f"in/s{c['seed']}_n{n}/target"]).to(DEV) if c["has_target"] else None
        want = json.loads(str(g1[key]))
        got = L.LogitsStatistics(L.LossConfig(**c["cfg"]))(q, cand, target)
        assert got.keys() == want.keys(), (key, got, want)
        for name in want:
            assert got[name] == pytest.approx(want[name], rel=2e-5, abs=2e-6), (key, name)


@pytest.mark.parametrize("k,H", [(1, 64), (7, 64), (50, 64), (7, 384)])  # 384: the reference's default d_model
@pytest.mark.parametrize("kind", OL.LOSS_KINDS)
def test_dense_hard_negatives_with_duplicate_candidates_vs_oracle(L, kind, k, H):
    """Duplicated candidate rows put exact ties at the k-th logit: the selected multiset (and so the loss and the
    gradient) must not depend on which of the equal candidates a top-k picks."""
    g = torch.Generator().manual_seed(11)
    N, C = 33, 300
    base = torch.randn(N, 40, H, generator=g)
    pick = torch.randint(0, 40, (N, C), generator=g)
    cand = torch.gather(base, 1, pick[..., None].expand(-1, -1, H)).contiguous()
    q = torch.randn(N, H, generator=g)
    cfg = dict(target_position="first", mask_false_negatives=True, num_hard_negatives=k, scale=3.0, margin=0.3)
    qo = q.clone().requires_grad_(True)
    want = OL.embed_loss(kind, qo, cand, None, **cfg)
    want.backward()
    qd = q.to(DEV).requires_grad_(True)
    got = getattr(L, kind)(L.LossConfig(**cfg))(qd, cand.to(DEV))
    assert got.item() == pytest.approx(want.item(), rel=2e-5, abs=1e-5)
    got.backward()
    _grad_close(qd.grad.cpu().numpy(), qo.grad.numpy(), f"{kind} k={k}")


def test_dense_loss_rejects_what_the_reference_rejects(L):
    q = torch.randn(4, 8, device=DEV)
    cand = torch.randn(4, 6, 8, device=DEV)
    with pytest.raises(AssertionError):  # losses.py:233-238: exactly one of target / target_position
        L.InfoNCELoss(L.LossConfig(target_position=None))(q, cand)
    with pytest.raises(AssertionError):
        L.InfoNCELoss(L.LossConfig(target_position="first"))(q, cand, torch.zeros(4, dtype=torch.long, device=DEV))
    with pytest.raises(AssertionError):  # losses.py:166-177
        L.InfoNCELoss(L.LossConfig())(q, cand[:3])


@pytest.mark.parametrize("cfg", [dict(), dict(mask_false_negatives=False, scale=4.0, margin=0.2),
                                 dict(target_position="diagonal"), dict(num_hard_negatives=5)])
@pytest.mark.parametrize("kind", OL.LOSS_KINDS)
def test_dense_loss_gradient_wrt_candidates_vs_oracle(L, kind, cfg):
    """EmbedLoss.forward is differentiable in candidate_embed too (losses.py:128-155; the cosine heads through the
    normalisation of the candidates, losses.py:196-208): d_candidates against the oracle's autograd, every head, dot and
    cosine logits, diagonal target, top-k restriction."""
    g = torch.Generator().manual_seed(21)
    N, C, H = 9, 23, 48
    q = torch.randn(N, H, generator=g)
    cand = torch.randn(N, C, H, generator=g) * torch.rand(N, C, 1, generator=g).add(0.3)  # assorted candidate norms
    full = dict(target_position="first", mask_false_negatives=True, num_hard_negatives=0, scale=1.0, margin=0.5) | cfg
    qo, co = q.clone().requires_grad_(True), cand.clone().requires_grad_(True)
    want = OL.embed_loss(kind, qo, co, None, **full)
    want.backward()
    qd, cd = q.to(DEV).requires_grad_(True), cand.to(DEV).requires_grad_(True)
    got = getattr(L, kind)(L.LossConfig(**full))(qd, cd)
    assert got.item() == pytest.approx(want.item(), rel=2e-5, abs=1e-5)
    (2.5 * got).backward()  # an upstream factor reaches both gradients
    _grad_close(qd.grad.cpu().numpy() / 2.5, qo.grad.numpy(), f"{kind} d_query")
    _grad_close(cd.grad.cpu().numpy().reshape(N * C, H) / 2.5, co.grad.numpy().reshape(N * C, H), f"{kind} d_candidates")
