"""The training step captured as a hipGraph (xfmr_rec_amd.trainer.GraphedStep): replay == eager, bit for bit, with dropout
ON -- the dropout stream and AdamW's step count are read from device memory inside the captured kernels
(xfmr_encoder_cfg.step_device, xfmr_adamw_dev, xfmr_step_advance), so every replay is a new step."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _setup(X, train_loss="InfoNCELoss", B=8, L=24, H=64, V=200, seed=0):
    g = torch.Generator().manual_seed(seed)
    table = torch.randn(V + 1, H, generator=g)
    table = table / table.norm(dim=-1, keepdim=True)
    table[0] = 0
    conf = X.LightningConfig(hidden_size=H, num_attention_heads=H // 32, intermediate_size=2 * H, num_hidden_layers=2,
                             max_seq_length=L, train_loss=train_loss)
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.set_table(table.to(DEV))
    batches = []
    for i in range(5):
        b = {k: torch.randint(1, V + 1, (B, L), generator=g) for k in ("history_item_idx", "pos_item_idx", "neg_item_idx")}
        n = 5 + 3 * i
        for k in b:
            b[k][1, n:] = 0  # one ragged row
        batches.append({k: v.to(DEV) for k, v in b.items()})
    return mod, batches


@pytest.fixture(scope="module")
def X():
    import xfmr_rec_amd as X

    return X


def _isolated(train_loss):
    """Child process of the cross-stream cases below (spawned: its own HIP context)."""
    import pathlib
    import sys

    root = pathlib.Path(__file__).resolve().parents[1]
    for p in (root, root / "transformer-recommenders_amd", root / "tests"):
        sys.path.insert(0, str(p))
    import xfmr_rec_amd as X

    _replay_equals_eager(X, train_loss, True)


@pytest.mark.parametrize("train_loss,overlap", [("InfoNCELoss", True), ("PairwiseLogisticLoss", True), ("InfoNCELoss", False)])
def test_graph_replay_equals_eager_steps_bit_for_bit_with_dropout(X, train_loss, overlap):
    """overlap=True captures the step WITH its two forks (weight-gradient GEMMs and logging heads on side streams: graph
    branches) -- not what the product replays (GraphedStep's default is the single-stream capture: cross-stream graphs replay
    2-3x slower on this runtime). Those two cases run in a spawned child: round 4 saw ONE segmentation fault inside
    hipGraphLaunch of such a graph, at its first replay, in a process that had run ~110 other GPU tests before it in an unusual
    order (full-size, two-rank, packed, then this file); not reproduced by any pair of files nor by the suite in its own
    order (three runs). A child keeps a runtime fault of the experimental form from taking the whole suite with it."""
    if not overlap:
        return _replay_equals_eager(X, train_loss, overlap)
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    proc = ctx.Process(target=_isolated, args=(train_loss,))
    proc.start()
    proc.join(300)
    assert proc.exitcode == 0, f"cross-stream capture / replay child exited with {proc.exitcode}"


def _replay_equals_eager(X, train_loss, overlap):
    # eager: the same device-side step counter drives dropout and AdamW
    eager, batches = _setup(X, train_loss)
    eager.model.use_device_step(True)
    tr_e = X.Trainer(eager)
    tr_e.optimizer.step_device = eager.model.step_device
    graphed, _ = _setup(X, train_loss)
    tr_g = X.Trainer(graphed)
    # 3 eager warm-up steps on batches[0]; the capture itself runs nothing. overlap: the captured step forks its weight-gradient
    # GEMMs and its logging heads onto side streams (graph branches) -- the eager reference runs everything in line
    step = X.GraphedStep(tr_g, batches[0], warmup=3, overlap=overlap)
    for _ in range(3):
        tr_e.fit_step(batches[0])
    torch.cuda.synchronize()
    assert int(eager.model.step_device) == int(graphed.model.step_device) == 3
    assert torch.equal(eager.model.flat, graphed.model.flat)
    losses_e, losses_g = [], []
    for b in batches[1:4]:  # three more steps: eager vs replay
        losses_e.append(tr_e.fit_step(b).clone())
        losses_g.append(step(b).clone())
        torch.cuda.synchronize()
        assert torch.equal(eager.model.flat, graphed.model.flat)
        assert torch.equal(eager.model.flat.grad, graphed.model.flat.grad)
    assert [float(x) for x in losses_e] == [float(x) for x in losses_g]
    # the masks did change from step to step: the same batch replayed twice gives different losses with dropout on
    a = float(step(batches[4]).clone())
    b = float(step(batches[4]).clone())
    assert a != b
    # the logged values are device tensors the replay refreshes
    assert set(step.logged) >= {f"loss/{c.__name__}" for c in X.LOSS_CLASSES} | {"batch/positive_density", "logits/neg/mean"}
    assert float(step.logged[f"loss/{train_loss}"]) == b


def test_device_step_counter_changes_the_dropout_mask_and_matches_adamw_by_value(X):
    """xfmr_adamw_dev(step read on the device) == xfmr_adamw(step by value), and the encoder forward with the same host
    seed but another counter value draws another mask."""
    from xfmr_rec_amd import ops

    n = 4099
    g = torch.Generator().manual_seed(1)
    p0, gr = torch.randn(n, generator=g).to(DEV), torch.randn(n, generator=g).to(DEV)
    for t in (1, 2, 7, 1000):
        pa, ma, va = p0.clone(), torch.full((n,), 0.01, device=DEV), torch.full((n,), 0.02, device=DEV)
        pb, mb, vb = p0.clone(), ma.clone(), va.clone()
        ops.adamw_(pa, gr, ma, va, lr=1e-3, step=t)
        cnt = torch.tensor([t - 1], dtype=torch.int32, device=DEV)
        ops.adamw_(pb, gr, mb, vb, lr=1e-3, step_device=cnt)
        assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb), t
        ops.step_advance_(cnt)
        assert int(cnt) == t
    mod, batches = _setup(X)
    mod.train()
    m = mod.model
    m.use_device_step(True)
    tok0, _ = m._encode_tokens(batches[0]["history_item_idx"])
    tok0b, _ = m._encode_tokens(batches[0]["history_item_idx"])
    assert torch.equal(tok0, tok0b)  # same counter -> same mask (the host-side step count no longer enters)
    ops.step_advance_(m.step_device)
    tok1, _ = m._encode_tokens(batches[0]["history_item_idx"])
    assert not torch.equal(tok0, tok1)


@pytest.mark.parametrize("mode,host", [("on", False), ("auto", False), ("on", True)])
def test_fit_with_graph_runs_the_same_training_as_eager_steps(X, mode, host):
    """Trainer.fit(graph=...): three eager steps, then (mode "on") every step a replay of ONE captured hipGraph, or (mode
    "auto") timed eager steps, timed replays and the faster form for the rest. Whatever ran, the sequence of steps is the
    eager one with the device-side step counter: same losses, same parameters, bit for bit; a batch of another shape
    takes the eager step; batches handed over from host memory (the ring's device slots) replay the same."""
    eager, batches = _setup(X)
    eager.model.use_device_step(True)
    tr_e = X.Trainer(eager)
    tr_e.optimizer.step_device = eager.model.step_device
    graphed, _ = _setup(X)
    tr_g = X.Trainer(graphed)
    seq = [batches[i % len(batches)] for i in range(16)]
    short = {k: v[:5].clone() for k, v in batches[2].items()}  # a short last batch
    seq.append(short)
    want = [float(tr_e.fit_step(b)) for b in seq]
    if host:  # batches arrive in host memory: PinnedBatchRing slots feed the captured buffers
        seq = [{k: v.cpu() for k, v in b.items()} for b in seq]
    got = tr_g.fit(seq, graph=mode, graph_probe_steps=4)
    assert got == want
    torch.cuda.synchronize()
    assert torch.equal(eager.model.flat, graphed.model.flat)
    assert int(graphed.model.step_device) == len(seq)
    if mode == "on":
        assert tr_g.graph_choice == "graph"
    else:
        assert tr_g.graph_choice in ("graph", "eager") and set(tr_g.graph_probe) == {"eager_ms", "graph_ms"}


def test_capture_after_default_stream_eager_steps_is_refused_with_a_python_error(X):
    """torch's capture protocol: eager steps on the device's DEFAULT stream bind autograd's AccumulateGrad node
    of `flat` to that stream; a capture on a side stream afterwards faulted in round 3. Now (1) the step keeps only
    DETACHED tensors past its end -- no autograd graph survives into the next step --, and (2) a Trainer that has run
    default-stream steps refuses `fit(graph=...)` / GraphedStep with a clear RuntimeError instead of attempting the capture.
    Steps on a side stream, or a fresh Trainer, capture as before."""
    mod, batches = _setup(X)
    tr = X.Trainer(mod)
    loss = tr.fit_step(batches[0])  # default stream
    assert tr.default_stream_steps == 1
    assert loss.grad_fn is None and all(not (torch.is_tensor(v) and v.requires_grad) for v in mod.last_out.values())
    with pytest.raises(RuntimeError, match="default stream"):
        tr.fit([batches[1], batches[2]], graph="on")
    with pytest.raises(RuntimeError, match="default stream"):
        X.GraphedStep(tr, batches[0])
    assert tr.fit([batches[1]], graph="off")  # eager steps go on
    # the same module under a NEW trainer whose eager steps run on a side stream: the capture is taken
    mod2, _ = _setup(X)
    tr2 = X.Trainer(mod2)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        tr2.fit_step(batches[0])
    torch.cuda.current_stream().wait_stream(side)
    assert tr2.default_stream_steps == 0
    out = tr2.fit([batches[i % 5] for i in range(6)], graph="on")
    assert len(out) == 6 and tr2.graph_choice == "graph"


def test_step_counter_is_one_sequence_across_eager_steps_replays_and_state_dict(X):
    """GraphedStep seeds the device-side counter from the optimizer's completed steps (AdamW's bias correction
    and the dropout stream continue, they do not restart at 1 on warm moments), and the replays' steps reach
    optimizer.state_dict() (they advance the counter on the device only)."""
    mod, batches = _setup(X)
    tr = X.Trainer(mod)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for i in range(4):
            tr.fit_step(batches[i])  # host-side step count: 4 completed steps
        gs = X.GraphedStep(tr, batches[0], warmup=0)
        torch.cuda.synchronize()
        assert int(mod.model.step_device.item()) == 4
        for i in range(3):
            gs(batches[i])
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    assert int(mod.model.step_device.item()) == 7
    sd = tr.optimizer.state_dict()
    assert all(st["step"] == 7 for st in sd["state"].values())
    # fit() restores the caller's defer_logging and leaves the counter written back
    mod3, _ = _setup(X)
    mod3.defer_logging = "auto"
    tr3 = X.Trainer(mod3)
    tr3.fit([batches[i % 5] for i in range(7)], graph="on")
    assert mod3.defer_logging == "auto"
    assert all(st["step"] == 7 for st in tr3.optimizer.state.values())


def test_deferred_step_logs_the_batch_constants_and_ring_needs_two_slots(X):
    """trainer.py:241-244 logs batch/size, batch/seq_len, batch/numel every step: the sync-free path carries them too (host
    constants, no device sync). PinnedBatchRing(slots=1) would stage into the slot the running step reads: refused."""
    from xfmr_rec_amd.data import PinnedBatchRing

    mod, batches = _setup(X)
    tr = X.Trainer(mod)
    tr.fit_step(batches[0])
    B, L = batches[0]["history_item_idx"].shape
    assert int(mod.logged["batch/size"]) == B and int(mod.logged["batch/seq_len"]) == L and int(mod.logged["batch/numel"]) == B * L
    vals = mod.logged_values(mod.last_out)
    assert vals["batch/size"] == B and vals["batch/numel"] == B * L
    assert vals["batch/attention_density"] == pytest.approx(float(mod.logged["batch/attention_density"]), rel=1e-6)
    with pytest.raises(ValueError, match="at least 2 slots"):
        PinnedBatchRing(DEV, B, L, slots=1)
    # a pageable batch staged over and over through two slots: every take returns what was staged
    ring = PinnedBatchRing(DEV, B, L, slots=2)
    host = [{k: v.cpu() + i for k, v in batches[0].items()} for i in range(5)]
    ring.stage(host[0])
    for i in range(5):
        got = ring.take()
        if i + 1 < 5:
            ring.stage(host[i + 1])
        for k in host[i]:
            assert torch.equal(got[k].cpu(), host[i][k]), (i, k)
    ring.release()
    ring.close()


def test_unit_gradient_backward_hook_skips_nothing_but_launches(X):
    """RecommenderLightningModule.backward (Lightning's hook) = loss.backward() with the persistent unit gradient: the fused
    loss's backward recognises the object (no `d_tok *= 1` launch, no ones-fill) and every gradient is bit-identical to a
    plain loss.backward(); any other upstream gradient still multiplies."""
    from xfmr_rec_amd import ops

    mod, batches = _setup(X)
    mod.train()
    res = []
    for how in ("plain", "hook", "scaled"):
        mod.model.flat.grad = None
        mod.model._step = 7  # the same dropout masks for the three runs
        loss = mod.training_step(batches[0], 0)
        hits = ops.unit_grad_hits
        if how == "plain":
            loss.backward()
        elif how == "hook":
            mod.backward(loss)
            assert ops.unit_grad_hits == hits + 1
        else:
            (2.0 * loss).backward()
        mod.on_train_batch_end(loss, batches[0], 0)
        res.append(mod.model.flat.grad.clone())
    assert torch.equal(res[0], res[1]) and float(res[0].abs().max()) > 0
    assert torch.equal(res[2], 2.0 * res[0])
