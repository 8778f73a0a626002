"""torch.ops.xfmr.* (xfmr_rec_amd/custom_ops.py): the hot path's kernels as registered PyTorch custom operators.

CPU half (no device): every op exists with a schema, its fake implementation propagates the shapes / dtypes the kernels
produce (FakeTensorMode, fake "cuda" tensors), and the differentiable ones are functional (a requirement of
register_autograd). GPU half: ``torch.library.opcheck`` on real inputs (schema, fake-vs-real, autograd registration, AOT
dispatch), the registered-op route equals the eager ``autograd.Function`` route bit for bit, and the model + loss trace
with ``torch.compile(fullgraph=True)`` -- no graph break -- to the same values and gradients.
"""

import ctypes

import pytest
import torch

from helpers import unit_table

DEV = "cuda"


@pytest.fixture(scope="module")
def X():
    import xfmr_rec_amd as pkg

    return pkg


def _enc_args(ops, *, heads=2, inter=128, layers=2, max_pos=24, precision="bf16", **kw):
    return ops.encoder_op_args(heads=heads, inter=inter, layers=layers, max_pos=max_pos, precision=precision, **kw)


def test_ops_are_registered_with_schemas_and_differentiable_ones_are_functional(X):
    from xfmr_rec_amd import custom_ops

    for name in custom_ops.OP_NAMES:
        op = getattr(torch.ops.xfmr, name).default
        assert op._schema.name == f"xfmr::{name}"
    for name in ("encoder", "sampled_loss", "sampled_loss_lists", "dense_loss", "l2_normalize"):
        assert not getattr(torch.ops.xfmr, name).default._schema.is_mutable, name
    s = str(torch.ops.xfmr.encoder.default._schema)
    assert "Tensor flat_params, Tensor item_idx, Tensor table" in s and "-> (Tensor, Tensor, Tensor)" in s
    assert "!) d_tok" in str(torch.ops.xfmr.encoder_bwd.default._schema)  # declared: the backward clobbers d_tok
    assert "!) params" in str(torch.ops.xfmr.adamw_.default._schema)


def test_fake_tensors_propagate_the_kernels_shapes_without_a_device(X):
    """register_fake of every op under FakeTensorMode with fake HIP tensors: what torch.compile / torch.export run to
    infer shapes. The activation workspace's size comes from the library's own host-side xfmr_encoder_workspace_bytes."""
    from torch._subclasses.fake_tensor import FakeTensorMode

    from xfmr_rec_amd import _native as N
    from xfmr_rec_amd import ops

    B, L, H, V = 4, 24, 64, 100
    cfg = ops.make_encoder_cfg(batch=B, seq_len=L, hidden=H, heads=2, inter=128, layers=2, max_pos=L, precision="bf16")
    n_params = N.load().xfmr_param_count(ctypes.byref(cfg))
    nbytes = N.load().xfmr_encoder_workspace_bytes(ctypes.byref(cfg))
    with FakeTensorMode():
        flat = torch.empty(n_params, device=DEV)
        idx = torch.empty(B, L, dtype=torch.int64, device=DEV)
        table = torch.empty(V + 1, H, device=DEV)
        tok, km, acts = torch.ops.xfmr.encoder(flat, idx, table, *_enc_args(ops))
        assert tok.shape == (B, L, H) and tok.dtype == torch.float32 and tok.device.type == "cuda"
        assert km.shape == (B, L) and km.dtype == torch.uint8
        assert acts.shape == (max(nbytes, 16),) and acts.dtype == torch.uint8
        g = torch.ops.xfmr.encoder_bwd(flat, torch.empty_like(tok), km, acts, *_enc_args(ops))
        assert g.shape == flat.shape
        rn = torch.empty(V + 1, device=DEV)
        pos = torch.empty(B * L, dtype=torch.int64, device=DEV)
        sc = (N.LOSS_IDS["InfoNCELoss"], 1, True, N.NEG_SHARED, 1.0, 0.5, "bf16", 0)
        loss, losses, stats, d = torch.ops.xfmr.sampled_loss(tok.view(B * L, H), km.view(-1), pos, pos, table, rn, None, *sc,
                                                              True, [0, 0])
        assert loss.shape == () and losses.shape == (14,) and stats.shape == (16,) and d.shape == (B * L, H)
        assert torch.ops.xfmr.sampled_loss(tok.view(B * L, H), km.view(-1), pos, pos, table, rn, None, *sc, False, [])[3].shape == (0,)
        q = torch.empty(7, H, device=DEV)
        out = torch.ops.xfmr.sampled_loss_lists(q, pos[:7], pos, table, rn, None, *sc, True)
        assert out[0].shape == () and out[3].shape == (7, H)
        cand = torch.empty(7, 9, H, device=DEV)
        out = torch.ops.xfmr.dense_loss(q, cand, None, N.TARGET_FIRST, 3, 0, True, 1.0, 0.5, 0, True, True)
        assert out[3].shape == (7, H) and out[4].shape == (7, 9, H)
        out = torch.ops.xfmr.dense_loss(q, cand, None, N.TARGET_FIRST, 3, 0, True, 1.0, 0.5, 0, True, False)
        assert out[4].shape == (0,)
        y, inv = torch.ops.xfmr.l2_normalize(q, 1e-12)
        assert y.shape == q.shape and inv.shape == (7,)
        assert torch.ops.xfmr.l2_normalize_bwd(y, y, inv, 1e-12).shape == q.shape
        assert torch.ops.xfmr.pool(tok, km, 0).shape == (B, H)
        assert torch.ops.xfmr.adamw_(flat, flat, flat, flat, 1e-3, 0.9, 0.999, 1e-8, 0.01, 1, 1.0, None) is None
        # an encoder the library does not build (head size 48) is refused at trace time, not at run time
        with pytest.raises(Exception):
            torch.ops.xfmr.encoder(flat, idx, torch.empty(V + 1, 96, device=DEV), *_enc_args(ops))


# ------------------------------------------------------------------------------------------------------------ GPU
def _module(X, prec="bf16", train_loss="InfoNCELoss", H=64, A=2, I=128, nL=2, L=24, V=100):
    conf = X.LightningConfig(hidden_size=H, num_attention_heads=A, intermediate_size=I, num_hidden_layers=nL,
                             max_seq_length=L, precision=prec, train_loss=train_loss)
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.set_table(unit_table(V, H).to(DEV))
    return mod


def _batch(B=4, L=24, V=100, seed=0):
    g = torch.Generator().manual_seed(seed)
    lengths = [L, 17, 5, L][:B]
    batch = {k: torch.zeros(B, L, dtype=torch.int64) for k in ("history_item_idx", "pos_item_idx", "neg_item_idx")}
    for b, n in enumerate(lengths):
        for k in batch:
            batch[k][b, :n] = torch.randint(1, V + 1, (n,), generator=g)
    return {k: v.to(DEV) for k, v in batch.items()}


@pytest.mark.gpu
def test_opcheck_on_the_device(X):
    """torch.library.opcheck: schema (declared mutations are the only ones), fake vs real outputs, the autograd
    registration, and the AOT-dispatch path, for the ops with a backward."""
    from xfmr_rec_amd import _native as N
    from xfmr_rec_amd import ops

    B, L, H, V = 4, 24, 64, 100
    cfg = ops.make_encoder_cfg(batch=B, seq_len=L, hidden=H, heads=2, inter=128, layers=2, max_pos=L, precision="bf16")
    g = torch.Generator().manual_seed(1)
    flat = (0.05 * torch.randn(N.load().xfmr_param_count(ctypes.byref(cfg)), generator=g)).to(DEV).requires_grad_(True)
    table = unit_table(V, H).to(DEV)
    idx = _batch()["history_item_idx"]
    utils = ("test_schema", "test_faketensor", "test_autograd_registration", "test_aot_dispatch_static")
    # (the encoder's third output is its activation WORKSPACE: raw bytes with uninitialised gaps between the carved tensors,
    #  so two runs differ there -- test_aot_dispatch_static compares outputs bit for bit and is left to the ops below and to
    #  test_model_and_loss_trace_without_a_graph_break, which runs the encoder under AOTAutograd and compares what matters)
    torch.library.opcheck(torch.ops.xfmr.encoder, (flat, idx, table, *_enc_args(ops)), test_utils=utils[:3])
    tok, km, _ = torch.ops.xfmr.encoder(flat.detach(), idx, table, *_enc_args(ops))
    rn, tb = ops.table_prepare(table)
    b = _batch()
    sc = (N.LOSS_IDS["InfoNCELoss"], 1, True, N.NEG_SHARED, 1.0, 0.5, "bf16", 0)
    t = tok.view(B * L, H).clone().requires_grad_(True)
    torch.library.opcheck(torch.ops.xfmr.sampled_loss,
                          (t, km.view(-1), b["pos_item_idx"].view(-1), b["neg_item_idx"].view(-1), table, rn, tb, *sc, True, [0, 0]),
                          test_utils=utils)
    q = torch.randn(7, H, generator=g).to(DEV).requires_grad_(True)
    cand = torch.randn(7, 9, H, generator=g).to(DEV).requires_grad_(True)
    torch.library.opcheck(torch.ops.xfmr.dense_loss, (q, cand, None, N.TARGET_FIRST, 3, 0, True, 1.0, 0.5, 0, True, True),
                          test_utils=utils)
    torch.library.opcheck(torch.ops.xfmr.l2_normalize, (q, 1e-12), test_utils=utils)
    torch.library.opcheck(torch.ops.xfmr.pool, (tok, km, 0), test_utils=("test_schema", "test_faketensor"))


@pytest.mark.gpu
@pytest.mark.parametrize("train_loss", ["InfoNCELoss", "AlignmentContrastiveLoss"])
def test_registered_ops_equal_the_eager_route_bit_for_bit(X, train_loss, monkeypatch):
    """XFMR_TORCH_OPS=1 sends the eager step through torch.ops.xfmr.* (dispatcher + registered autograd formulas)
    instead of the autograd.Functions: same C-ABI calls on the same data -- every loss and every gradient identical."""
    batch = _batch()
    res = []
    for route in ("0", "1"):
        monkeypatch.setenv("XFMR_TORCH_OPS", route)
        mod = _module(X, train_loss=train_loss)
        mod.eval()
        out = mod.compute_losses(batch, sync_metrics=False)
        out[f"loss/{train_loss}"].backward()
        res.append(({k: v.detach().clone() for k, v in out.items() if k.startswith("loss/")}, mod.model.flat.grad.clone()))
    for k in res[0][0]:
        assert torch.equal(res[0][0][k], res[1][0][k]), k
    assert torch.equal(res[0][1], res[1][1]) and float(res[0][1].abs().max()) > 0
    # the reference's calling convention, EmbedLoss.forward(query, dense candidates) + LossConfig variants
    from xfmr_rec_amd import losses as XL

    g = torch.Generator().manual_seed(3)
    q0, cand0 = torch.randn(9, 32, generator=g).to(DEV), torch.randn(9, 11, 32, generator=g).to(DEV)
    outs = []
    for route in ("0", "1"):
        monkeypatch.setenv("XFMR_TORCH_OPS", route)
        q, cand = q0.clone().requires_grad_(True), cand0.clone().requires_grad_(True)
        loss = XL.PairwiseLogisticLoss(XL.LossConfig(target_position="diagonal", margin=0.2), precision="fp32")(q, cand)
        (2.5 * loss).backward()
        outs.append((loss.detach(), q.grad, cand.grad))
    assert all(torch.equal(a, b) for a, b in zip(*outs))


@pytest.mark.gpu
@pytest.mark.parametrize("train_loss", ["InfoNCELoss", "PairwiseLogisticLoss"])
def test_model_and_loss_trace_without_a_graph_break(X, train_loss):
    """torch.compile(fullgraph=True) over RecommenderModel.forward and over compute_losses (encoder + fused loss of all
    seven heads): dynamo must not break the graph on the kernels (they are opaque registered ops with fake
    implementations), AOTAutograd must trace their backward (registered autograd formulas), and values / gradients equal
    the eager step's. backend="aot_eager": the traced graph runs the same kernels (no code generator is involved -- and
    none is installed on the box)."""
    batch = _batch()
    mod = _module(X, train_loss=train_loss)
    mod.eval()
    want = mod.compute_losses(batch, sync_metrics=False)
    want[f"loss/{train_loss}"].backward()
    want_grad = mod.model.flat.grad.clone()
    mod.model.flat.grad = None
    torch._dynamo.reset()
    fwd = torch.compile(lambda idx: mod.model(idx), backend="aot_eager", fullgraph=True)
    with torch.no_grad():
        got_fwd, ref_fwd = fwd(batch["history_item_idx"]), mod.model(batch["history_item_idx"])
    for k in ("token_embeddings", "sentence_embedding", "attention_mask"):
        assert torch.equal(got_fwd[k], ref_fwd[k]), k
    step = torch.compile(lambda b: mod.compute_losses(b, sync_metrics=False), backend="aot_eager", fullgraph=True)
    got = step(batch)
    for k, v in want.items():
        if k.startswith("loss/"):
            assert torch.equal(got[k].detach(), v.detach()), k
    got[f"loss/{train_loss}"].backward()
    assert torch.equal(mod.model.flat.grad, want_grad)
    # a TRAINING step (dropout on) traces once the dropout stream is keyed on the device
    mod.train()
    mod.model.context()
    mod.model.use_device_step(True)
    torch._dynamo.reset()
    tstep = torch.compile(lambda b: mod.compute_losses(b, sync_metrics=False, defer_logging=False), backend="aot_eager",
                          fullgraph=True)
    mod.model.flat.grad = None
    a = tstep(batch)[f"loss/{train_loss}"]
    a.backward()
    g1 = mod.model.flat.grad.clone()
    mod.model.flat.grad = None
    b = mod.compute_losses(batch, sync_metrics=False, defer_logging=False)[f"loss/{train_loss}"]  # eager, same counter value
    b.backward()
    assert torch.equal(a.detach(), b.detach()) and torch.equal(g1, mod.model.flat.grad)
