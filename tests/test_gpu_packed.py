"""The PACKED layout (xfmr_encoder_cfg.seq_offsets, ABI 3): the encoder and the loss run each sequence's own rows -- the
reference right-pads every row to the batch's longest (data.py:799-805), computes the padding rows too and drops them
afterwards (models.py:392). Valid rows must get the padded layout's values: bit for bit where a row's arithmetic does not
depend on its neighbours (every forward kernel, the loss, the dX chain), to fp32 summation order where rows are summed
(weight / bias / LayerNorm-parameter gradients). And the CPU oracle (padded, as the reference) is the checker of the whole
step."""

import ctypes

import pytest
import torch

from helpers import TOL, grad_tol, loss_tol, rel_l2, unit_table

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def X():
    import xfmr_rec_amd as pkg

    return pkg


@pytest.fixture(scope="module")
def ops():
    from xfmr_rec_amd import ops as o

    return o


def _ragged(B, L, V, lengths, seed=0, hole=None):
    g = torch.Generator().manual_seed(seed)
    batch = {k: torch.zeros(B, L, dtype=torch.int64) for k in ("history_item_idx", "pos_item_idx", "neg_item_idx")}
    for b, n in enumerate(lengths):
        for k in batch:
            batch[k][b, :n] = torch.randint(1, V + 1, (n,), generator=g)
    if hole is not None:
        batch["pos_item_idx"][hole] = 0  # a valid position whose positive is padding (models.py:413)
    return batch


def _offsets(lengths):
    off = torch.zeros(len(lengths) + 1, dtype=torch.int64)
    off[1:] = torch.cumsum(torch.tensor(lengths, dtype=torch.int64), 0)
    return off


def test_pack_rows_kernel(ops):
    B, L, V = 7, 33, 50
    lengths = [33, 0, 1, 17, 32, 0, 5]
    batch = _ragged(B, L, V, lengths, seed=3)
    off = _offsets(lengths)
    rows = int(off[-1])
    pk = ops.pack_rows(batch["history_item_idx"].to(DEV), batch["pos_item_idx"].to(DEV), batch["neg_item_idx"].to(DEV),
                       off.to(DEV), rows)
    valid = torch.arange(L)[None, :] < torch.tensor(lengths)[:, None]
    for k, name in (("hist", "history_item_idx"), ("pos", "pos_item_idx"), ("neg", "neg_item_idx")):
        assert torch.equal(pk[k].cpu(), batch[name][valid])
    assert torch.equal(pk["seq_offsets"].cpu(), off.to(torch.int32))
    assert torch.equal(pk["row_pos"].cpu().long(), torch.arange(L)[None, :].expand(B, L)[valid])
    pk2 = ops.pack_rows(batch["history_item_idx"].to(DEV), batch["pos_item_idx"].to(DEV), None, off.to(DEV), rows)
    assert pk2["neg"] is None and torch.equal(pk2["hist"], pk["hist"])
    # in another order (xfmr_pack_rows_ordered): slot b holds batch row order[b] -- the same as packing the reordered batch
    order, off_o = ops.length_order(lengths)
    assert [lengths[i] for i in order.tolist()] == sorted(lengths, reverse=True)
    pk_o = ops.pack_rows(*(batch[k].to(DEV) for k in ("history_item_idx", "pos_item_idx", "neg_item_idx")), off_o.to(DEV), rows,
                         order=order.to(DEV))
    pk_r = ops.pack_rows(*(batch[k][order].contiguous().to(DEV) for k in ("history_item_idx", "pos_item_idx", "neg_item_idx")),
                         off_o.to(DEV), rows)
    for k in ("hist", "pos", "neg", "seq_offsets", "row_pos"):
        assert torch.equal(pk_o[k], pk_r[k]), k


@pytest.mark.parametrize("B,L,H,A,I,nL,lengths", [
    (6, 200, 128, 4, 512, 2, [200, 131, 17, 0, 1, 64]),          # config 2's layer shape; an empty and a one-row sequence
    (5, 40, 64, 2, 128, 2, [40, 23, 3, 40, 8]),
    (120, 200, 128, 4, 512, 2, None),                            # 24 000 padded rows: the LayerNorm-fused / fused-FFN kernels
    (3, 256, 256, 8, 1024, 1, [256, 100, 31]),                   # H = 256, the lock-step backward's longest sequence
    (3, 512, 128, 4, 512, 1, [512, 300, 31]),                    # 256 < L <= 512: four-tile forward deal, two-role backward
    (128, 32, 64, 2, 128, 2, "short"),                           # the e2e shape: 4 096 padded rows of H = 64, short sequences
    (64, 32, 64, 2, 128, 2, "short"),
])
def test_packed_encoder_equals_the_padded_layout_on_valid_rows(ops, B, L, H, A, I, nL, lengths):
    from xfmr_rec_amd import _native as N

    V = 300
    g = torch.Generator().manual_seed(11)
    if lengths == "short":
        lengths = torch.randint(11, L + 1, (B,), generator=g).tolist()
    elif lengths is None:
        lengths = torch.exp(4.35 + torch.randn(B, generator=g)).round().clamp(16, L).long().tolist()
    table = unit_table(V, H).to(DEV)
    batch = _ragged(B, L, V, lengths, seed=5)
    idx = batch["history_item_idx"].to(DEV)
    off = _offsets(lengths)
    rows = int(off[-1])
    pk = ops.pack_rows(idx, batch["pos_item_idx"].to(DEV), None, off.to(DEV), rows)
    kw = dict(batch=B, seq_len=L, hidden=H, heads=A, inter=I, layers=nL, max_pos=L, precision="bf16")
    cfg_pad = ops.make_encoder_cfg(**kw)
    cfg_pk = ops.make_encoder_cfg(**kw, seq_offsets=pk["seq_offsets"], row_pos=pk["row_pos"])
    n_params = N.load().xfmr_param_count(ctypes.byref(cfg_pad))
    flat = (0.05 * torch.randn(n_params, generator=g)).to(DEV)
    valid = (torch.arange(L)[None, :] < torch.tensor(lengths)[:, None]).to(DEV)
    tok_p, km_p, acts_p = ops.encoder_fwd(cfg_pad, flat, idx, table)
    tok_k, km_k, acts_k = ops.encoder_fwd(cfg_pk, flat, pk["hist"], table)
    assert tok_k.shape == (rows, H) and km_k.shape == (rows,)
    assert torch.equal(tok_k, tok_p[valid]) and torch.equal(km_k, km_p[valid])
    d_out = torch.randn(B, L, H, generator=g).to(DEV) * valid[..., None]  # padding rows carry no gradient (models.py:392)
    g_p = ops.encoder_bwd(cfg_pad, flat, d_out.clone(), km_p, acts_p)
    g_k = ops.encoder_bwd(cfg_pk, flat, d_out[valid].contiguous(), km_k, acts_k)
    assert torch.isfinite(g_k).all()
    assert rel_l2(g_k, g_p) <= 2e-5, rel_l2(g_k, g_p)
    g_k2 = ops.encoder_bwd(cfg_pk, flat, d_out[valid].contiguous(), km_k, acts_k)
    assert torch.equal(g_k, g_k2)  # bit-reproducible


def test_packed_layout_is_refused_where_no_kernel_walks_offsets(ops):
    from xfmr_rec_amd import _native as N

    B, L, H = 2, 16, 64
    off = _offsets([16, 5]).to(DEV)
    idx = torch.ones(B, L, dtype=torch.int64, device=DEV)
    pk = ops.pack_rows(idx, idx, None, off, 21)
    table = unit_table(20, H).to(DEV)
    for bad in (dict(precision="fp32"), dict(causal=False), dict(heads=1)):  # fp32 policy, bidirectional, head size 64
        kw = dict(batch=B, seq_len=L, hidden=H, heads=2, inter=128, layers=1, max_pos=L, precision="bf16") | bad
        cfg = ops.make_encoder_cfg(**kw, seq_offsets=pk["seq_offsets"], row_pos=pk["row_pos"])
        n = N.load().xfmr_param_count(ctypes.byref(cfg))
        assert N.load().xfmr_encoder_workspace_bytes(ctypes.byref(cfg)) == 0  # refused on the host
        with pytest.raises(RuntimeError):
            ops.encoder_fwd(cfg, torch.zeros(max(n, 1), device=DEV), pk["hist"], table)


def _module(X, H=64, A=2, I=128, nL=2, L=24, V=100, prec="bf16", train_loss="InfoNCELoss", **cfg):
    conf = X.LightningConfig(hidden_size=H, num_attention_heads=A, intermediate_size=I, num_hidden_layers=nL,
                             max_seq_length=L, precision=prec, train_loss=train_loss, **cfg)
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.set_table(unit_table(V, H).to(DEV))
    return mod


@pytest.mark.parametrize("train_loss,cfg", [("InfoNCELoss", {}), ("PairwiseLogisticLoss", {}), ("AlignmentContrastiveLoss", {}),
                                            ("InfoNCELoss", dict(negatives="catalogue", target_position=None,
                                                                 mask_false_negatives=False))])
def test_packed_training_step_equals_the_padded_step_and_the_oracle(X, train_loss, cfg):
    """compute_losses on a batch that carries its rows' lengths (as PinnedBatchRing hands them over) runs packed: every loss
    and statistic equals the padded step bit for bit, the gradient to summation order -- and both agree with the CPU
    oracle, which is padded like the reference."""
    from oracle import model as OM

    B, L, V, H, A = 6, 24, 100, 64, 2
    lengths = [24, 17, 5, 0, 1, 9]
    batch = _ragged(B, L, V, lengths, seed=1, hole=(1, 3))
    res = []
    for with_lengths in (False, True):
        mod = _module(X, train_loss=train_loss, **cfg)
        mod.eval()  # dropout off: its masks are keyed by the row index, which the layouts do not share
        b = {k: v.to(DEV) for k, v in batch.items()}
        if with_lengths:
            b["lengths"] = torch.tensor(lengths)
        out = mod.compute_losses(b)
        out[f"loss/{train_loss}"].backward()
        res.append((out, mod.model.flat.grad.clone(), mod))
    (o0, g0, _), (o1, g1, mod) = res
    for k, v in o0.items():
        if torch.is_tensor(v) and v.dim() == 0 or isinstance(v, (int, float)):
            assert float(o1[k]) == float(v) or (float(v) != float(v) and float(o1[k]) != float(o1[k])), (k, float(v), float(o1[k]))
    assert rel_l2(g1, g0) <= 2e-5 and float(g0.abs().max()) > 0
    if not cfg:  # the oracle's in-batch form (materialised candidates, as the reference)
        params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mod.model.encoder_state_dict().items()}
        want = OM.compute_losses(params, unit_table(V, H), batch, num_heads=A, max_seq_length=L, loss_cfg={}, resolve_ties=True)
        want[f"loss/{train_loss}"].backward()
        for cls in X.LOSS_CLASSES:
            k = f"loss/{cls.__name__}"
            w = float(want[k].detach())
            assert abs(float(o1[k]) - w) <= loss_tol("bf16", w, flips=True), (k, float(o1[k]), w)
        for k in ("batch/attention_density", "batch/positive_density", "batch/numel"):
            assert float(o1[k]) == pytest.approx(float(want[k]), rel=1e-5), k
        got = mod.model.grad_state_dict()
        for k, p_ in params.items():
            if not k.endswith("key.bias"):
                assert rel_l2(got[k], p_.grad) <= grad_tol("bf16", flips=True), k


def test_packed_step_through_the_ring_trainer_and_the_registered_ops(X, monkeypatch):
    """PinnedBatchRing hands the rows' lengths over with every batch; Trainer.fit then trains on packed rows (the pack
    kernel runs once per step), fp32 / long-sequence models keep the padded layout, and the registered-op route
    (XFMR_TORCH_OPS=1) carries the offsets as tensors."""
    from xfmr_rec_amd import ops
    from xfmr_rec_amd.data import PinnedBatchRing, row_lengths

    B, L, V = 8, 24, 100
    lengths = [24, 3, 11, 24, 0, 7, 19, 2]
    host = [_ragged(B, L, V, lengths, seed=s) for s in range(4)]
    assert row_lengths(host[0]["history_item_idx"]).tolist() == lengths
    ring = PinnedBatchRing(DEV, B, L, slots=2)
    ring.stage(host[0])
    got = ring.take()
    assert got["lengths"].tolist() == lengths and got["packed_rows"] == sum(lengths)
    torch.cuda.synchronize()
    # the packed layout takes the rows longest first (ops.length_order): the ring ships that order and the offsets in it
    order = got["order"].cpu().tolist()
    assert sorted(order) == list(range(B)) and [lengths[i] for i in order] == sorted(lengths, reverse=True)
    assert order == [0, 3, 6, 2, 5, 1, 7, 4]  # (stable: equal lengths keep their batch order)
    assert got["offsets"].cpu().tolist() == _offsets([lengths[i] for i in order]).tolist()
    ring.release()
    ring.close()
    calls = []
    real = ops.pack_rows
    monkeypatch.setattr(ops, "pack_rows", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    mod = _module(X)
    tr = X.Trainer(mod)
    losses = tr.fit(host, graph="off")
    assert len(calls) == 4 and all(l == l and l > 0 for l in losses)
    # fp32 parity policy: no kernel walks offsets there -- the lengths are ignored, the padded layout runs
    calls.clear()
    mod32 = _module(X, prec="fp32")
    X.Trainer(mod32).fit(host[:2], graph="off")
    assert not calls
    # registered ops: same values as the autograd.Function route on a packed batch
    outs = []
    for route in ("0", "1"):
        monkeypatch.setenv("XFMR_TORCH_OPS", route)
        m = _module(X)
        m.eval()
        b = {k: v.to(DEV) for k, v in host[1].items()} | {"lengths": torch.tensor(lengths)}
        out = m.compute_losses(b, sync_metrics=False)
        out["loss/InfoNCELoss"].backward()
        outs.append((out["loss/InfoNCELoss"].detach().clone(), m.model.flat.grad.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
