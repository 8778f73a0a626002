"""GPU parity of the composed path (encoder, fused loss, training step) against the golden vectors produced
by the reference (tests/golden, see oracle/make_golden.py) and against the CPU oracle on larger seeded
inputs. Everything goes through the product API (xfmr_rec_amd) and therefore through libxfmr_hip.so."""

import json

import numpy as np
import pytest
import torch

from helpers import TOL, assert_close, grad_tol, loss_tol, ragged_batch, rel_l2, unit_table

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _t(a):
    return torch.from_numpy(np.asarray(a))


@pytest.fixture(scope="module")
def X():
    import xfmr_rec_amd as x

    return x


def _model(X, *, H, A, I, nL, Lmax, prec, state=None, table=None, is_decoder=True):
    cfg = X.ModelConfig(hidden_size=H, num_attention_heads=A, intermediate_size=I, num_hidden_layers=nL,
                        max_seq_length=Lmax, is_decoder=is_decoder)
    m = X.RecommenderModel(cfg, device=DEV, precision=prec)
    if state is not None:
        m.load_encoder_state_dict(state)
    if table is not None:
        m.set_table(table.to(DEV))
    return m.eval()


# ------------------------------------------------------------------------------------------ encoder (G2)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_encoder_matches_hf_bert_golden(X, golden_dir, case, prec):
    _encoder_vs_golden(X, np.load(golden_dir / "g2_encoder.npz"), case, prec, is_decoder=True)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_bidirectional_encoder_matches_hf_bert_golden(X, golden_dir, case, prec):
    """ModelConfig(is_decoder=False) (models.py:50,355): HF BertModel then attends in both directions; outputs and every
    parameter gradient against G5 (generated from transformers' BertModel(is_decoder=False), oracle/make_golden.py)."""
    _encoder_vs_golden(X, np.load(golden_dir / "g5_encoder_bidirectional.npz"), case, prec, is_decoder=False)


def _encoder_vs_golden(X, g2, case, prec, is_decoder):
    cfg = json.loads(str(g2[f"{case}/cfg"]))
    pre = f"{case}/param/"
    state = {k[len(pre):]: _t(g2[k]) for k in g2.files if k.startswith(pre)}
    m = _model(X, H=cfg["H"], A=cfg["A"], I=cfg["I"], nL=cfg["nL"], Lmax=cfg["L"], prec=prec, state=state,
               table=torch.zeros(2, cfg["H"]), is_decoder=is_decoder)
    x = _t(g2[f"{case}/x"]).to(DEV)
    mask = _t(g2[f"{case}/mask"]).bool()
    out = m(item_embeds=x)
    assert torch.equal(out["attention_mask"].cpu().bool(), mask)
    tok = out["token_embeddings"]
    assert_close("last_hidden_state", tok.cpu()[mask], _t(g2[f"{case}/eager/last_hidden_state"])[mask], prec)
    from oracle import encoder as enc

    assert_close("sentence_embedding", out["sentence_embedding"],
                 enc.mean_pool(_t(g2[f"{case}/eager/last_hidden_state"]), mask), prec)
    w = torch.linspace(0.5, 1.5, cfg["H"], device=DEV)
    loss = ((tok * w) ** 2 * mask.to(DEV)[..., None]).sum()
    loss.backward()
    grads = m.grad_state_dict()
    gp = f"{case}/eager/grad/"
    worst = 0.0
    for k in g2.files:
        if k.startswith(gp):
            name = k[len(gp):]
            if name.endswith("key.bias"):  # exactly zero in exact arithmetic: pure rounding noise
                assert grads[name].abs().max().item() < 1e-2
                continue
            worst = max(worst, assert_close(name, grads[name], _t(g2[k]), prec, "grad"))
    assert worst > 0.0


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_encoder_config2_shape_vs_oracle(X, prec):
    """One real-size layer stack (L=200, H=128, 4 heads, I=512) on ragged rows against the CPU oracle."""
    from oracle import encoder as enc
    from oracle import model as OM

    B, L, H, A, I, nL, V = 3, 200, 128, 4, 512, 2, 300
    table = unit_table(V, H)
    batch, lengths = ragged_batch(B, L, V, lengths=[200, 131, 17], seed=2)
    m = _model(X, H=H, A=A, I=I, nL=nL, Lmax=L, prec=prec, table=table)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.encoder_state_dict().items()}
    ref = OM.forward(params, table, batch["history_item_idx"], num_heads=A, max_seq_length=L)
    out = m(batch["history_item_idx"].to(DEV))
    valid = ref["attention_mask"].bool()
    assert torch.equal(out["attention_mask"].cpu().bool(), valid)
    assert_close("tok", out["token_embeddings"].cpu()[valid], ref["token_embeddings"].detach()[valid], prec)
    w = torch.linspace(-1, 1, H)
    (ref["token_embeddings"] * w * valid[..., None]).sum().backward()
    (out["token_embeddings"] * w.to(DEV) * valid.to(DEV)[..., None]).sum().backward()
    got = m.grad_state_dict()
    for k, p in params.items():
        if k.endswith("key.bias"):
            continue
        assert_close(k, got[k], p.grad, prec, "grad")
    assert enc.num_layers_of(params) == nL


@pytest.mark.parametrize("prec,L,lengths", [("fp32", 200, [200, 131, 17]), ("bf16", 200, [200, 131, 17])])
def test_encoder_config4_shape_vs_oracle(X, prec, L, lengths):
    """BASELINE config 4's layer shape (L=200, H=256, 8 heads, I=1024; 2 of its 6 layers) on ragged rows against the
    CPU oracle: every trainable tensor's gradient. These shapes take other code than config 2: no LayerNorm-fused GEMM
    epilogues (H != 128), ln_*_v4_kernel<64 lanes per row>, 64x128 forward tiles for N = 768 / 1024, K = 1024 dX."""
    _encoder_shape_vs_oracle(X, prec, B=3, L=L, H=256, A=8, I=1024, nL=2, V=300, lengths=lengths)


@pytest.mark.parametrize("prec,L,lengths", [("bf16", 512, [512, 300, 5]), ("fp32", 256, [256, 131, 1])])
def test_encoder_config5_shape_vs_oracle(X, prec, L, lengths):
    """BASELINE config 5's layer shape (H=256, 8 heads, I=1024) at L=512 in bf16 (two-block attention forward and the
    dQ + dK/dV backward pair: the fused one-workgroup forms stop at L=256) and at L=256 in the fp32 parity policy
    (its attention keeps whole fp32 panels in LDS: L <= 256)."""
    _encoder_shape_vs_oracle(X, prec, B=3, L=L, H=256, A=8, I=1024, nL=2, V=300, lengths=lengths)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("H,A,I", [(512, 16, 256), (320, 10, 160), (384, 6, 96), (768, 12, 192), (128, 2, 256)])
def test_encoder_other_d_model_vs_oracle(X, prec, H, A, I):
    """d_model outside {64, 128, 256, 384} (any (V, H) table) and head size 64 (models.py:22-48 takes
    any hidden_size / num_attention_heads pair): 512 / 16 and 320 / 10 heads (head size 32), 384 / 6, 768 / 12 (the
    all-mpnet shape) and 128 / 2 (head size 64) through gather, LayerNorms, GEMMs and attention against the CPU oracle,
    every trainable tensor's gradient."""
    _encoder_shape_vs_oracle(X, prec, B=3, L=40, H=H, A=A, I=I, nL=2, V=150, lengths=[40, 23, 3])


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("H,A", [(512, 16), (320, 10), (384, 6)])
@pytest.mark.parametrize("train_loss", ["InfoNCELoss", "AlignmentContrastiveLoss"])
def test_training_step_other_d_model_vs_oracle(X, prec, H, A, train_loss):
    """The whole step -- encoder, the fused loss of all seven heads, backward -- at a d_model the LDS-DMA loss kernel is not
    instantiated for: the generic loss kernel (next template width, masked rows) in both precision policies, with and
    without the bf16 table copy, against the oracle's materialised form."""
    from oracle import model as OM

    L, V, B, I = 24, 90, 4, 128
    table = unit_table(V, H)
    batch, lengths = ragged_batch(B, L, V, lengths=[24, 17, 5, 24], seed=5)
    conf = X.LightningConfig(hidden_size=H, num_attention_heads=A, intermediate_size=I, num_hidden_layers=1,
                             max_seq_length=L, precision=prec, train_loss=train_loss)
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.set_table(table.to(DEV))
    mod.eval()
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in mod.model.encoder_state_dict().items()}
    want = OM.compute_losses(params, table, batch, num_heads=A, max_seq_length=L, loss_cfg={}, resolve_ties=True)
    want[f"loss/{train_loss}"].backward()
    out = mod.compute_losses(batch)
    out[f"loss/{train_loss}"].backward()
    tol = TOL[prec]
    for cls in X.LOSS_CLASSES:
        k = f"loss/{cls.__name__}"
        w = float(want[k].detach())
        assert abs(float(out[k]) - w) <= loss_tol(prec, w, flips=True), (k, float(out[k]), w)  # default cfg: masked
    got = mod.model.grad_state_dict()
    for k, p_ in params.items():
        if k.endswith("key.bias"):
            continue
        e = rel_l2(got[k], p_.grad)
        assert e <= grad_tol(prec, flips=True), (k, e)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_encoder_reference_default_shape_vs_oracle(X, prec):
    """The reference's own default model (config.yaml:46-52 + models.py:80-91 with the all-MiniLM table): d_model 384,
    12 heads, intermediate 48, 1 layer, max_seq_length 32. K = 48 takes the 32-deep K slices."""
    _encoder_shape_vs_oracle(X, prec, B=5, L=32, H=384, A=12, I=48, nL=1, V=200, lengths=[32, 31, 17, 2, 1])


@pytest.mark.parametrize("prec,L,lengths", [("fp32", 200, [200, 131, 17]), ("bf16", 200, [200, 131, 17]),
                                            ("bf16", 300, [300, 129, 2])])
def test_bidirectional_encoder_config2_shape_vs_oracle(X, prec, L, lengths):
    """is_decoder=False at the bench's layer shape (H=128, 4 heads, I=512; 2 layers): two and three 128-row blocks per
    sequence, every block reading ALL keys (the causal kernels stop at the diagonal), ragged rows."""
    _encoder_shape_vs_oracle(X, prec, B=3, L=L, H=128, A=4, I=512, nL=2, V=300, lengths=lengths, is_decoder=False)


def _encoder_shape_vs_oracle(X, prec, *, B, L, H, A, I, nL, V, lengths, is_decoder=True):
    from oracle import model as OM

    table = unit_table(V, H)
    batch, lengths = ragged_batch(B, L, V, lengths=lengths, seed=2)
    m = _model(X, H=H, A=A, I=I, nL=nL, Lmax=L, prec=prec, table=table, is_decoder=is_decoder)
    params = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in m.encoder_state_dict().items()}
    ref = OM.forward(params, table, batch["history_item_idx"], num_heads=A, max_seq_length=L, causal=is_decoder)
    out = m(batch["history_item_idx"].to(DEV))
    valid = ref["attention_mask"].bool()
    assert torch.equal(out["attention_mask"].cpu().bool(), valid)
    assert_close("tok", out["token_embeddings"].cpu()[valid], ref["token_embeddings"].detach()[valid], prec)
    assert_close("sentence_embedding", out["sentence_embedding"], ref["sentence_embedding"].detach(), prec)
    w = torch.linspace(-1, 1, H)
    (ref["token_embeddings"] * w * valid[..., None]).sum().backward()
    (out["token_embeddings"] * w.to(DEV) * valid.to(DEV)[..., None]).sum().backward()
    got = m.grad_state_dict()
    for k, p in params.items():
        if k.endswith("key.bias"):
            continue
        assert_close(k, got[k], p.grad, prec, "grad")


@pytest.mark.parametrize("mode,normalized", [("mean", False), ("max", True), ("cls", True), ("lasttoken", True)])
def test_encode_drops_unknown_ids_and_pooling_normalize_through_forward(X, mode, normalized):
    """`encode(item_ids)` (models.py:347-364): ids -> rows through id2idx, unknown ids dropped, the pooled embedding of
    the remaining history; `forward` applies `pooling_mode` and, with `is_normalized`, the Normalize module
    (models.py:143-147) -- against the oracle's encoder + pooling on the same rows. (max / cls / lasttoken follow the
    oracle's restatement of sentence-transformers Pooling: parity unpinned for those modes)"""
    import torch.nn.functional as F

    from oracle import encoder as enc
    from oracle import model as OM

    H, A, I, nL, L, V = 64, 2, 96, 2, 16, 30
    g = torch.Generator().manual_seed(0)
    emb = torch.randn(V, H, generator=g)
    ids = [f"item-{i}" for i in range(V)]
    cfg = X.ModelConfig(hidden_size=H, num_attention_heads=A, intermediate_size=I, num_hidden_layers=nL, max_seq_length=L,
                        pooling_mode=mode, is_normalized=normalized)
    m = X.RecommenderModel(cfg, device=DEV, precision="fp32").eval()
    m.configure_embeddings({"item_id": ids, "embedding": emb.numpy()})
    assert m.embeddings.is_cuda and m.table_rnorm is not None and torch.all(m.embeddings[0] == 0)
    history = ["item-3", "no-such-item", "item-17", "item-3", "???", "item-29"]
    got = m.encode(history)
    rows = torch.tensor([[4, 18, 4, 30]])  # item i of the dataset is row i + 1; the two unknown ids are gone
    assert torch.equal(got, m(rows.to(DEV))["sentence_embedding"][0])
    params = {k: v.detach().cpu() for k, v in m.encoder_state_dict().items()}
    table = torch.cat([torch.zeros(1, H), emb])
    ref = OM.forward(params, table, rows, num_heads=A, max_seq_length=L)
    want = enc.pool(ref["token_embeddings"], ref["attention_mask"], mode)[0]
    if normalized:
        want = F.normalize(want, p=2, dim=0)
    assert_close(f"encode[{mode}]", got, want, "fp32")
    # a longer history than max_seq_length keeps its last L items (models.py:334-337)
    long_hist = [f"item-{i % V}" for i in range(40)]
    assert torch.equal(m.encode(long_hist), m.encode(long_hist[-L:]))


def test_truncation_to_max_seq_length(X):
    """models.py:334-337: only the last max_seq_length items are encoded."""
    H, V = 64, 40
    table = unit_table(V, H)
    m = _model(X, H=H, A=2, I=64, nL=1, Lmax=8, prec="fp32", table=table)
    idx = torch.randint(1, V + 1, (2, 13), generator=torch.Generator().manual_seed(0))
    a = m(idx.to(DEV))["token_embeddings"]
    b = m(idx[:, -8:].to(DEV))["token_embeddings"]
    assert a.shape == (2, 8, H) and torch.equal(a, b)


# ------------------------------------------------------------------------------------------ loss heads (G4)
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_loss_heads_match_reference_golden(X, golden_dir, prec):
    from xfmr_rec_amd import losses as XL
    from xfmr_rec_amd import ops

    g4 = np.load(golden_dir / "g4_shared_negatives.npz")
    table = _t(g4["table"]).to(DEV)
    rnorm = ops.table_rnorm(table)
    q0, pos, neg = _t(g4["q"]).to(DEV), _t(g4["pos"]).to(DEV), _t(g4["neg"]).to(DEV)
    classes = {c.__name__: c for c in XL.LOSS_CLASSES}
    for case in json.loads(str(g4["index"])):
        cfg = XL.LossConfig(**case["cfg"])
        q = q0.clone().requires_grad_(True)
        cand = XL.SharedNegatives(table, rnorm, pos, neg)
        assert tuple(cand.shape) == (q.shape[0], 1 + neg.numel(), q.shape[1])
        loss = classes[case["kind"]](cfg, precision=prec)(q, cand)
        loss.backward()
        want = float(g4[f"{case['key']}/loss"])
        # bf16 logits flip a few false-negative mask bits: 1e-2 relative (+ small abs floor)
        flips = bool(cfg.mask_false_negatives) or cfg.num_hard_negatives > 0
        assert abs(loss.item() - want) <= loss_tol(prec, want, flips=flips), (case["key"], loss.item(), want)
        e = rel_l2(q.grad, _t(g4[f"{case['key']}/dq"]))
        assert e <= grad_tol(prec, flips=flips), (case["key"], e)
    # exact ties (a sampled negative IS the row's positive): the kernel resolves them by item id == the
    # mathematical `logits < pos_logit`; the reference's outcome depends on its bmm rounding and is recorded
    # in the fixture. Kernel vs the oracle with id-resolved ties; oracle vs reference where the reference
    # happened to drop every tie.
    from oracle import losses as OL

    neg_t = _t(g4["ties/neg"])
    tab_c, pos_c, q_c = _t(g4["table"]), _t(g4["pos"]), _t(g4["q"])
    cand_c = torch.cat([tab_c[pos_c][:, None], tab_c[neg_t][None].expand(q_c.shape[0], -1, -1)], 1)
    ties = torch.cat([torch.zeros(q_c.shape[0], 1, dtype=torch.bool), neg_t[None] == pos_c[:, None]], 1)
    assert int(ties.sum()) == 3
    for cls in XL.LOSS_CLASSES:
        name = cls.__name__
        want = float(OL.embed_loss(name, q_c, cand_c, ties=ties))
        if not g4[f"ties/{name}/tie_counted_as_negative"].any():
            assert want == pytest.approx(float(g4[f"ties/{name}/loss"]), rel=1e-5, abs=1e-5)
        got = cls(XL.LossConfig(), precision=prec)(q0, XL.SharedNegatives(table, rnorm, pos, neg_t.to(DEV)))
        assert abs(float(got) - want) <= loss_tol(prec, want, flips=True), name  # LossConfig(): masked
    for vi in range(5):
        want = json.loads(str(g4[f"v{vi}/stats"]))
        cfgd = [c["cfg"] for c in json.loads(str(g4["index"])) if c["key"].startswith(f"v{vi}/")][0]
        got = XL.LogitsStatistics(XL.LossConfig(**cfgd), precision=prec)(q0, XL.SharedNegatives(table, rnorm, pos, neg))
        assert got.keys() == want.keys()
        for k, v in want.items():
            assert got[k] == pytest.approx(v, rel=3e-2 if prec == "bf16" else 2e-4, abs=2e-2 if prec == "bf16" else 1e-5), k


@pytest.mark.parametrize("mask_fn", [True, False])
@pytest.mark.parametrize("k", [1, 5, 40])
def test_hard_negatives_on_structured_candidates_vs_oracle(X, golden_dir, k, mask_fn):
    """losses.py:295-330 on the fused path: per-row top-k thresholds over never-materialised logits. Negatives
    repeat items (ties at the threshold), the oracle materialises the (Np,1+N,H) tensor and runs torch.topk."""
    from oracle import losses as OL
    from xfmr_rec_amd import losses as XL
    from xfmr_rec_amd import ops

    g4 = np.load(golden_dir / "g4_shared_negatives.npz")
    tab_c, pos_c, q_c = _t(g4["table"]), _t(g4["pos"]), _t(g4["q"])
    gen = torch.Generator().manual_seed(5)
    neg_c = torch.randint(101, 201, (150,), generator=gen)  # 100 distinct items drawn 150 times: duplicates
    table = tab_c.to(DEV)
    rnorm = ops.table_rnorm(table)
    cfgd = dict(mask_false_negatives=mask_fn, num_hard_negatives=k, scale=4.0, margin=0.4)
    cand_c = torch.cat([tab_c[pos_c][:, None], tab_c[neg_c][None].expand(q_c.shape[0], -1, -1)], 1)
    for cls in XL.LOSS_CLASSES:
        name = cls.__name__
        qo = q_c.clone().requires_grad_(True)
        want = OL.embed_loss(name, qo, cand_c, **cfgd)
        want.backward()
        q = q_c.to(DEV).requires_grad_(True)
        got = cls(XL.LossConfig(**cfgd), precision="fp32")(q, XL.SharedNegatives(table, rnorm, pos_c.to(DEV), neg_c.to(DEV)))
        assert abs(got.item() - want.item()) <= TOL["fp32"]["loss_rel"] * max(1.0, abs(want.item())), (name, got.item(), want.item())
        got.backward()
        assert rel_l2(q.grad, qo.grad) <= TOL["fp32"]["grad_l2"], name
    want = OL.logits_statistics(q_c, cand_c, **cfgd)
    got = XL.LogitsStatistics(XL.LossConfig(**cfgd), precision="fp32")(
        q_c.to(DEV), XL.SharedNegatives(table, rnorm, pos_c.to(DEV), neg_c.to(DEV)))
    assert got.keys() == want.keys()
    for key, v in want.items():
        assert got[key] == pytest.approx(v, rel=2e-4, abs=1e-5), key
    # bf16 MFMA policy: the selected set may differ by elements at the threshold -> loose tolerance
    for cls in (XL.InfoNCELoss, XL.PairwiseLogisticLoss, XL.AlignmentContrastiveLoss):
        want = OL.embed_loss(cls.__name__, q_c, cand_c, **cfgd).item()
        got = cls(XL.LossConfig(**cfgd), precision="bf16")(
            q_c.to(DEV), XL.SharedNegatives(table, rnorm, pos_c.to(DEV), neg_c.to(DEV))).item()
        assert abs(got - want) <= loss_tol("bf16", want, flips=True), (cls.__name__, got, want)  # top-k selection


def test_hard_negatives_full_catalogue_vs_oracle(X, golden_dir):
    from oracle import losses as OL
    from xfmr_rec_amd import losses as XL
    from xfmr_rec_amd import ops

    g4 = np.load(golden_dir / "g4_shared_negatives.npz")
    tab_c, pos_c, q_c = _t(g4["table"]), _t(g4["pos"]), _t(g4["q"])
    table = tab_c.to(DEV)
    rnorm = ops.table_rnorm(table)
    cfgd = dict(target_position=None, mask_false_negatives=False, num_hard_negatives=7)
    cand_c = tab_c[None].expand(q_c.shape[0], -1, -1)
    for cls in XL.LOSS_CLASSES:
        qo = q_c.clone().requires_grad_(True)
        want = OL.embed_loss(cls.__name__, qo, cand_c, pos_c, **cfgd)
        want.backward()
        q = q_c.to(DEV).requires_grad_(True)
        got = cls(XL.LossConfig(**cfgd), precision="fp32")(q, XL.CatalogCandidates(table, rnorm, q.shape[0]), pos_c.to(DEV))
        assert abs(got.item() - want.item()) <= TOL["fp32"]["loss_rel"] * max(1.0, abs(want.item())), cls.__name__
        got.backward()
        assert rel_l2(q.grad, qo.grad) <= TOL["fp32"]["grad_l2"], cls.__name__


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_full_catalogue_softmax_matches_reference_golden(X, golden_dir, prec):
    """EmbedLoss.forward(q, table[None].expand, target=pos) == CE(Q E^T) (BASELINE config 4 mode)."""
    from xfmr_rec_amd import losses as XL
    from xfmr_rec_amd import ops

    g4 = np.load(golden_dir / "g4_shared_negatives.npz")
    table = _t(g4["table"]).to(DEV)
    rnorm = ops.table_rnorm(table)
    q0, pos = _t(g4["q"]).to(DEV), _t(g4["pos"]).to(DEV)
    cfg = XL.LossConfig(target_position=None, mask_false_negatives=False)
    for cls in XL.LOSS_CLASSES:
        q = q0.clone().requires_grad_(True)
        loss = cls(cfg, precision=prec)(q, XL.CatalogCandidates(table, rnorm, q.shape[0]), pos)
        loss.backward()
        want = float(g4[f"catalog/{cls.__name__}/loss"])
        assert abs(loss.item() - want) <= TOL[prec]["loss_rel"] * max(1.0, abs(want)), (cls.__name__, loss.item(), want)
        assert rel_l2(q.grad, _t(g4[f"catalog/{cls.__name__}/dq"])) <= TOL[prec]["grad_l2"], cls.__name__


def test_loss_api_errors_mirror_reference(X):
    from xfmr_rec_amd import losses as XL

    table = unit_table(10, 64).to(DEV)
    from xfmr_rec_amd import ops

    rn = ops.table_rnorm(table)
    cand = XL.SharedNegatives(table, rn, torch.ones(4, dtype=torch.int64, device=DEV), torch.ones(6, dtype=torch.int64, device=DEV))
    fn = XL.InfoNCELoss(XL.LossConfig())
    with pytest.raises(AssertionError):
        fn(torch.zeros(4, 1, 64, device=DEV), cand)  # query must be 2-D (losses.py:166)
    with pytest.raises(AssertionError):
        fn(torch.zeros(5, 64, device=DEV), cand)  # batch mismatch (losses.py:172)
    with pytest.raises(AssertionError):
        fn(torch.zeros(4, 64, device=DEV), cand, torch.zeros(4, dtype=torch.int64, device=DEV))  # both target and position
    # dense candidates that require a gradient get one (losses.py:128-155 is differentiable in both arguments;
    # values are checked in tests/test_gpu_dense_loss.py)
    dc = torch.randn(4, 7, 64, device=DEV, requires_grad=True)
    fn(torch.randn(4, 64, device=DEV), dc).backward()
    assert dc.grad is not None and dc.grad.shape == dc.shape and torch.isfinite(dc.grad).all()


# ------------------------------------------------------------------------------------------ fused loss, positions form
@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("T_shape,H,V", [((4, 50), 64, 60), ((5, 200), 128, 500), ((2, 96), 256, 100),
                                         ((6, 32), 384, 150),  # 384 = the reference's default d_model (all-MiniLM table)
                                         # any other d_model (models.py:22-48 takes any hidden_size): the generic kernel at
                                         # the next instantiated width, rows masked beyond their real width
                                         ((3, 40), 512, 120), ((2, 33), 320, 90), ((2, 40), 96, 70), ((2, 24), 640, 80),
                                         ((2, 20), 1024, 70)])
def test_fused_loss_positions_form_vs_oracle(X, prec, T_shape, H, V):
    """All seven heads + statistics + d_tok from ONE launch sequence, on ragged positions, vs the oracle that
    materialises the (Np, 1+N, H) candidates like models.py:408-416. Several tiles and splits are exercised."""
    from oracle import losses as OL
    from xfmr_rec_amd import _native as N
    from xfmr_rec_amd import ops

    if prec == "fp32" and H > 512:
        pytest.skip("fp32 parity policy: the fused loss stops at d_model 512 (its fp32 tile image must fit LDS)")
    B, L = T_shape
    table = unit_table(V, H)
    batch, lengths = ragged_batch(B, L, V, seed=3)
    g = torch.Generator().manual_seed(9)
    tok = torch.randn(B, L, H, generator=g) * 1.2
    key_mask = (batch["history_item_idx"] != 0)
    am = key_mask
    pos_i = batch["pos_item_idx"][am]
    keep = pos_i != 0
    tdev, rn = table.to(DEV), None
    rn = ops.table_rnorm(tdev)
    for head in OL.LOSS_KINDS:
        tq = tok.clone().requires_grad_(True)
        q = tq[am][keep]
        cand = torch.cat([table[pos_i[keep]][:, None], table[batch["neg_item_idx"][am]][None].expand(int(keep.sum()), -1, -1)], 1)
        ties = torch.cat([torch.zeros(q.shape[0], 1, dtype=torch.bool),
                          batch["neg_item_idx"][am][None, :] == pos_i[keep][:, None]], 1)  # resolved by item id
        want = {k: OL.embed_loss(k, q, cand, ties=ties) for k in OL.LOSS_KINDS}
        want[head].backward()
        losses, stats, d_tok = ops.sampled_loss(
            tok.to(DEV), key_mask.to(torch.uint8).to(DEV), batch["pos_item_idx"].to(DEV), batch["neg_item_idx"].to(DEV),
            tdev, rn, train_head=head, all_heads=True, precision=prec,
        )
        s = stats.tolist()
        assert int(s[N.STAT["n_valid"]]) == int(am.sum()) and int(s[N.STAT["n_query"]]) == int(keep.sum())
        for i, k in enumerate(OL.LOSS_KINDS):
            w = float(want[k])
            lim = loss_tol(prec, w, flips=True)  # default LossConfig: mask_false_negatives=True
            assert abs(losses[i].item() - w) <= lim, (head, k, losses[i].item(), w)
        e = rel_l2(d_tok, tq.grad)
        lim = grad_tol(prec, flips=True)
        if prec == "bf16" and head == "PairwiseHingeLoss":
            lim = 0.25  # the hinge sub-gradient is an indicator of (l_ij > c_i): bf16 logits flip it near the kink
        assert e <= lim, (head, e)
        # lean mode (only the train head) must give the same loss and the same gradient
        l2, _s2, d2 = ops.sampled_loss(
            tok.to(DEV), key_mask.to(torch.uint8).to(DEV), batch["pos_item_idx"].to(DEV), batch["neg_item_idx"].to(DEV),
            tdev, rn, train_head=head, all_heads=False, precision=prec,
        )
        i = OL.LOSS_KINDS.index(head)
        assert abs(l2[i].item() - losses[i].item()) <= 1e-5 * max(1.0, abs(losses[i].item()))
        assert rel_l2(d2, d_tok) <= 1e-5
    ref_stats = OL.logits_statistics(tok[am][keep], cand.detach(), ties=ties)
    got = X.losses.stats_to_dict(s)
    for k, v in ref_stats.items():
        assert got[k] == pytest.approx(v, rel=3e-2 if prec == "bf16" else 2e-4, abs=2e-2 if prec == "bf16" else 1e-5), k


# ------------------------------------------------------------------------------------------ training step (G3)
def _g3_module(X, golden_dir, prec, train_loss):
    g3 = np.load(golden_dir / "g3_step.npz")
    cfg = json.loads(str(g3["cfg"]))
    conf = X.LightningConfig(hidden_size=cfg["H"], num_attention_heads=cfg["A"], intermediate_size=cfg["I"],
                             num_hidden_layers=cfg["nL"], max_seq_length=cfg["L"], train_loss=train_loss,
                             precision=prec)
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.load_encoder_state_dict({k[len("param0/"):]: _t(g3[k]) for k in g3.files if k.startswith("param0/")})
    mod.model.set_table(_t(g3["table"]).to(DEV))
    batch = {"history_item_idx": _t(g3["hist"]), "pos_item_idx": _t(g3["pos"]), "neg_item_idx": _t(g3["neg"])}
    return g3, mod.to(DEV), batch


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_compute_losses_matches_reference_golden(X, golden_dir, prec):
    g3, mod, batch = _g3_module(X, golden_dir, prec, "InfoNCELoss")
    mod.eval()
    out = mod.compute_losses(batch)
    for cls in X.LOSS_CLASSES:
        k = f"loss/{cls.__name__}"
        want = float(g3[k])
        assert abs(float(out[k]) - want) <= loss_tol(prec, want, flips=True), (k, float(out[k]), want)  # masked (default)
        assert float(out[k + "Mean"]) == pytest.approx(float(out[k]) / (out["batch/positive_non_zero"] + 1e-9), rel=1e-5)
    assert out["batch/attention_non_zero"] == int(g3["attention_mask"].sum())
    assert out["batch/positive_non_zero"] == int(g3["positive_mask"].sum())
    for k, v in json.loads(str(g3["stats"])).items():
        assert out[k] == pytest.approx(v, rel=3e-2 if prec == "bf16" else 2e-4, abs=2e-2 if prec == "bf16" else 1e-5), k
    # the drop-in compute_embeds: same valid/positive masks and query rows as the reference
    e = mod.model.compute_embeds(batch["history_item_idx"], batch["pos_item_idx"], batch["neg_item_idx"])
    assert torch.equal(e["attention_mask"].cpu(), _t(g3["attention_mask"]))
    assert torch.equal(e["positive_mask"].cpu(), _t(g3["positive_mask"]))
    assert_close("query_embed", e["query_embed"], _t(g3["query_embed"]), prec)


@pytest.mark.parametrize("train_loss", ["InfoNCELoss", "AlignmentContrastiveLoss"])
def test_normalized_queries_and_hard_negatives_in_the_training_step_vs_oracle(X, golden_dir, train_loss):
    """ModelConfig.is_normalized (models.py:393-394) and LossConfig.num_hard_negatives (losses.py:295-330) through
    compute_losses + backward, on the G3 inputs, against the CPU oracle (fp32 policy, ties resolved by item id)."""
    from oracle import model as OM

    g3 = np.load(golden_dir / "g3_step.npz")
    cfg = json.loads(str(g3["cfg"]))
    conf = X.LightningConfig(hidden_size=cfg["H"], num_attention_heads=cfg["A"], intermediate_size=cfg["I"],
                             num_hidden_layers=cfg["nL"], max_seq_length=cfg["L"], train_loss=train_loss,
                             precision="fp32", is_normalized=True, num_hard_negatives=4, scale=3.0)
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    state = {k[len("param0/"):]: _t(g3[k]) for k in g3.files if k.startswith("param0/")}
    mod.model.load_encoder_state_dict(state)
    mod.model.set_table(_t(g3["table"]).to(DEV))
    mod = mod.to(DEV).eval()
    batch = {"history_item_idx": _t(g3["hist"]), "pos_item_idx": _t(g3["pos"]), "neg_item_idx": _t(g3["neg"])}
    out = mod.compute_losses(batch)
    loss_cfg = dict(target_position="first", mask_false_negatives=True, num_hard_negatives=4, scale=3.0, margin=0.5)
    params = {k: v.clone().requires_grad_(True) for k, v in state.items()}
    want = OM.compute_losses(params, _t(g3["table"]), batch, num_heads=cfg["A"], max_seq_length=cfg["L"],
                             loss_cfg=loss_cfg, is_normalized=True, resolve_ties=True)
    for cls in X.LOSS_CLASSES:
        k = f"loss/{cls.__name__}"
        w = float(want[k])
        assert abs(float(out[k]) - w) <= TOL["fp32"]["loss_rel"] * max(1.0, abs(w)), (k, float(out[k]), w)
    for k in ("logits/neg/density", "logits/neg/mean", "logits/pos/mean", "logits/neg/max"):
        assert out[k] == pytest.approx(want[k], rel=2e-4, abs=1e-5), k
    out[f"loss/{train_loss}"].backward()
    want[f"loss/{train_loss}"].backward()
    from xfmr_rec_amd.models import flat_layout

    flat_g = mod.model.flat.grad
    names, shapes, offsets, _total = flat_layout(cfg["H"], cfg["I"], cfg["L"], cfg["nL"])
    checked = 0
    for name, shape, off in zip(names, shapes, offsets):
        if name not in params or "key.bias" in name:  # d/d(key.bias) is exactly 0: rounding noise only
            continue
        n = int(np.prod(shape))
        g = flat_g[off:off + n].view(shape).cpu()
        assert rel_l2(g, params[name].grad) <= 5e-4, name
        checked += 1
    assert checked >= 10


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("train_loss", ["InfoNCELoss", "PairwiseLogisticLoss", "AlignmentContrastiveLoss"])
def test_reference_default_lightning_config_trains(X, prec, train_loss):
    """The reference's own defaults (config.yaml:46-52, models.py:37-48, 80-91; params.py:11 all-MiniLM-L6-v2 item
    embeddings => d_model 384, no projection): 12 heads, intermediate 48, 1 layer, max_seq_length 32, batch 32. Only
    `hidden_size` is given (the reference resolves it to 384 by downloading the model). Three optimizer steps against the
    CPU oracle of the same step: every head's value, the train head's parameter gradients, the parameters afterwards."""
    from oracle import model as OM

    conf = X.LightningConfig(hidden_size=384, train_loss=train_loss, precision=prec)
    assert (conf.num_attention_heads, conf.intermediate_size, conf.num_hidden_layers, conf.max_seq_length) == (12, 48, 1, 32)
    V, B, L, H = 500, 32, 32, 384
    table = unit_table(V, H)
    g = torch.Generator().manual_seed(4)
    lengths = torch.randint(1, L + 1, (B,), generator=g).tolist()
    lengths[0] = L
    batch, _ = ragged_batch(B, L, V, lengths=lengths, seed=5)
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.set_table(table.to(DEV))
    mod.eval()  # dropout off: parity is only defined without it
    tr = OM.OracleTrainer({k: v.detach().cpu().clone() for k, v in mod.model.encoder_state_dict().items()}, table,
                          num_heads=12, max_seq_length=L, train_loss=train_loss)
    opt = mod.configure_optimizers()
    tol = TOL[prec]
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        tr.opt.zero_grad(set_to_none=True)
        want = OM.compute_losses(tr.params, table, batch, num_heads=12, max_seq_length=L, loss_cfg={}, resolve_ties=True)
        want[f"loss/{train_loss}"].backward()
        loss = mod.training_step(batch)
        loss.backward()
        mod.on_train_batch_end(loss, batch, 0)  # (Lightning's hook: joins the logging stream, logs the other heads)
        for cls in X.LOSS_CLASSES:
            k = f"loss/{cls.__name__}"
            w = float(want[k].detach())
            assert abs(float(mod.logged[k]) - w) <= loss_tol(prec, w, flips=True), (step, k, float(mod.logged[k]), w)
        got = mod.model.grad_state_dict()
        for k, p_ in tr.params.items():
            if k.endswith("key.bias"):
                continue
            e = rel_l2(got[k], p_.grad)
            assert e <= grad_tol(prec, flips=True), (step, k, e)  # masked (default LossConfig)
        opt.step()
        tr.opt.step()
    if prec == "fp32":
        # three AdamW steps of lr 1e-3: early Adam moves every element by ~lr * sign(g), so an element whose gradient is
        # rounding noise around 0 may differ by a fraction of a step; 1e-4 = 3 % of the total movement
        for k, v in mod.model.encoder_state_dict().items():
            if k.endswith("key.bias"):  # gradient exactly 0 in exact arithmetic: Adam follows the sign of rounding noise
                continue
            assert (v.cpu() - tr.params[k].detach()).abs().max().item() <= 1e-4, k


@pytest.mark.parametrize("train_loss", ["InfoNCELoss", "PairwiseLogisticLoss", "AlignmentContrastiveLoss"])
def test_three_adamw_steps_match_reference_golden(X, golden_dir, train_loss):
    """zero_grad -> training_step -> backward -> AdamW, three times, fp32 MFMA path, dropout off (eval):
    gradients after step 1 and parameters after step 3 against the reference's own run."""
    g3, mod, batch = _g3_module(X, golden_dir, "fp32", train_loss)
    mod.eval()
    opt = mod.configure_optimizers()
    for step in range(3):
        opt.zero_grad(set_to_none=True)
        loss = mod.training_step(batch)
        assert float(loss) == pytest.approx(float(g3[f"{train_loss}/loss_step{step}"]), rel=2e-4)
        loss.backward()
        if step == 0 and f"{train_loss}/grad0/embeddings.LayerNorm.weight" in g3.files:
            for k, v in mod.model.grad_state_dict().items():
                if not k.endswith("key.bias"):
                    assert_close(k, v, _t(g3[f"{train_loss}/grad0/{k}"]), "fp32", "grad")
        opt.step()
    for k, v in mod.model.encoder_state_dict().items():
        atol = 3.5e-3 if k.endswith("key.bias") else 3e-5
        torch.testing.assert_close(v.cpu(), _t(g3[f"{train_loss}/param_after3/{k}"]), rtol=2e-4, atol=atol, msg=k)
    sd = mod.state_dict()
    assert "model.embeddings.weight" not in sd and any(k.startswith("model.model.0.auto_model.") for k in sd)


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
def test_full_catalogue_training_step_equals_dense_cross_entropy(X, golden_dir, prec):
    """LightningConfig.negatives='catalogue' (BASELINE config 4): the training step's InfoNCE is
    cross_entropy(Q E^T, pos, 'sum') over every row of the item table -- the reference API's
    EmbedLoss.forward(q, table[None].expand(N,-1,-1), target=pos_idx), target_position=None, mask_false_negatives
    False -- and its parameter gradients are those of that dense form pushed through the encoder backward."""
    g3 = np.load(golden_dir / "g3_step.npz")
    cfg = json.loads(str(g3["cfg"]))
    conf = X.LightningConfig(hidden_size=cfg["H"], num_attention_heads=cfg["A"], intermediate_size=cfg["I"],
                             num_hidden_layers=cfg["nL"], max_seq_length=cfg["L"], train_loss="InfoNCELoss",
                             precision=prec, negatives="catalogue", target_position=None,
                             mask_false_negatives=False)  # bf16: logging pass first, gradient pass with pinned maximum
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.load_encoder_state_dict({k[len("param0/"):]: _t(g3[k]) for k in g3.files if k.startswith("param0/")})
    mod.model.set_table(_t(g3["table"]).to(DEV))
    mod = mod.to(DEV).eval()
    batch = {"history_item_idx": _t(g3["hist"]), "pos_item_idx": _t(g3["pos"])}  # no negatives are sampled
    mod.model.flat.requires_grad_(True)
    out = mod.compute_losses(batch)
    out["loss/InfoNCELoss"].backward()
    got_grad = mod.model.flat.grad.clone()
    # the closed form on the same token embeddings (torch on the device as the checker)
    mod.model.flat.grad = None
    tok, key_mask = mod.model._encode_tokens(batch["history_item_idx"])
    pos = batch["pos_item_idx"].to(DEV)[:, -tok.shape[1]:]
    rows = key_mask.bool() & (pos != 0)
    want = torch.nn.functional.cross_entropy(tok[rows] @ mod.model.embeddings.T, pos[rows], reduction="sum")
    want.backward()
    assert float(out["loss/InfoNCELoss"]) == pytest.approx(float(want), rel=TOL[prec]["loss_rel"])
    assert rel_l2(got_grad, mod.model.flat.grad) <= (5e-4 if prec == "fp32" else TOL[prec]["grad_l2"])
    # the lean (train-head-only) evaluation takes the other kernel path (no logging pass): same value and gradient
    lean = X.RecommenderLightningModule(X.LightningConfig(**(conf.model_dump() | {"log_all_losses": False})))
    lean.configure_model()
    lean.model.load_encoder_state_dict({k[len("param0/"):]: _t(g3[k]) for k in g3.files if k.startswith("param0/")})
    lean.model.set_table(_t(g3["table"]).to(DEV))
    lean = lean.to(DEV).eval()
    lean.model.flat.requires_grad_(True)
    out2 = lean.compute_losses(batch)
    out2["loss/InfoNCELoss"].backward()
    assert float(out2["loss/InfoNCELoss"]) == pytest.approx(float(out["loss/InfoNCELoss"]), rel=1e-5)
    assert rel_l2(lean.model.flat.grad, got_grad) <= 1e-4
    assert out["batch/positive_non_zero"] == int(rows.sum())
    with pytest.raises(ValueError):  # the catalogue form names the positive by `target`
        bad = X.LightningConfig(**(conf.model_dump() | {"target_position": "first"}))
        m2 = X.RecommenderLightningModule(bad)
        m2.configure_model()
        m2.model.set_table(_t(g3["table"]).to(DEV))
        m2.to(DEV).compute_losses(batch)


def test_training_mode_dropout_and_trainer_loop(X, golden_dir):
    """Training mode (dropout 0.1 as TF:configuration_bert.py): loss differs from eval, is finite, decreases."""
    # (InfoNCE with false-negative masking is not monotone even in the reference's own run -- g3 loss_step0..2
    #  rise -- because fewer negatives stay masked as the model improves; BPR decreases.)
    g3, mod, batch = _g3_module(X, golden_dir, "bf16", "PairwiseLogisticLoss")
    key = "loss/PairwiseLogisticLoss"
    mod.eval()
    before = float(mod.compute_losses(batch)[key])
    tr = X.Trainer(mod)
    losses = tr.fit([batch] * 20)
    assert all(np.isfinite(losses))
    mod.eval()
    e1 = float(mod.compute_losses(batch)[key])
    e2 = float(mod.compute_losses(batch)[key])
    assert e1 == e2  # eval is deterministic
    assert e1 < before, (before, e1)
    mod.train()
    t1 = float(mod.compute_losses(batch)[key])
    assert t1 != e1  # dropout is live in training mode


def test_save_load_roundtrip(X, golden_dir, tmp_path):
    g3, mod, batch = _g3_module(X, golden_dir, "fp32", "InfoNCELoss")
    mod.eval()
    mod.save(tmp_path / "m")
    m2 = X.RecommenderModel.load(str(tmp_path / "m"), device=DEV, precision="fp32").eval()
    m2.set_table(mod.model.embeddings)
    a = mod.model(batch["history_item_idx"].to(DEV))["sentence_embedding"]
    b = m2(batch["history_item_idx"].to(DEV))["sentence_embedding"]
    assert torch.equal(a, b)


def test_deferred_logging_on_side_stream_matches_inline(X, golden_dir):
    """compute_losses(defer_logging=True) evaluates the logging heads on a side stream; after sync_logging() the
    values are bit-identical to the single-stream evaluation."""
    g3, mod, batch = _g3_module(X, golden_dir, "bf16", "InfoNCELoss")
    mod.eval()
    mod.model.flat.requires_grad_(True)
    a = mod.compute_losses(batch, sync_metrics=False)
    b = mod.compute_losses(batch, sync_metrics=False, defer_logging=True)
    b["loss/InfoNCELoss"].backward()
    mod.sync_logging()
    torch.cuda.synchronize()
    assert torch.equal(a["stats/device"], b["stats/device"])
    for i, cls in enumerate(X.LOSS_CLASSES):
        if cls.__name__ == "InfoNCELoss":  # the side-stream pass skips the train head (all_heads = 2)
            assert float(b["losses/device"][i]) == 0.0
        else:
            assert float(a[f"loss/{cls.__name__}"]) == float(b["losses/device"][i]), cls.__name__
    assert float(a["loss/InfoNCELoss"]) == float(b["loss/InfoNCELoss"])
    vals = mod.logged_values(b)
    for cls in X.LOSS_CLASSES:
        assert vals[f"loss/{cls.__name__}"] == float(a[f"loss/{cls.__name__}"])
        assert vals[f"loss/{cls.__name__}Mean"] == pytest.approx(float(a[f"loss/{cls.__name__}Mean"]), rel=1e-6)
    assert "logits/neg/density" in vals
