"""The product's data-parallel step with two ranks: `Trainer.fit_step` itself -- the HIP forward /
loss / backward on each rank's own shard (negatives stay rank-local), ONE SUM all-reduce of the flat gradient, the 1/W
folded into the fused AdamW launch -- must equal "average the gradients of W independent single-rank steps, then
AdamW" (what torch DDP does for the reference, config.yaml:5-6,35).

Only one GPU is at hand on the test box, so both ranks run on device 0 and the collective goes over gloo
(XFMR_REHEARSE_ONE_GPU=1; RCCL refuses two ranks on one device): what is exercised is everything of the N > 1 path
except RCCL's transport, which the driver's multi-GPU bench runs."""

import os
import pathlib
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = pathlib.Path(__file__).resolve().parents[1]
pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, out_dir, train_loss, single=False):
    for p in (ROOT, ROOT / "transformer-recommenders_amd", ROOT / "tests"):
        sys.path.insert(0, str(p))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port), XFMR_REHEARSE_ONE_GPU="1")
    import xfmr_rec_amd as X
    from helpers import ragged_batch, unit_table
    from xfmr_rec_amd import distributed as D

    r, local, w = D.init_process_group_from_env()
    assert (r, local, w) == (rank, 0, world)
    torch.cuda.set_device(0)
    H, A, I, nL, L, V, B = 64, 2, 128, 2, 24, 60, 8
    conf = X.LightningConfig(hidden_size=H, num_attention_heads=A, intermediate_size=I, num_hidden_layers=nL,
                             max_seq_length=L, train_loss=train_loss, precision="fp32")
    mod = X.RecommenderLightningModule(conf)
    mod.model = X.RecommenderModel(conf, device="cuda:0", precision="fp32", seed=3)  # same seed: replicated weights
    mod.configure_model()
    mod.model.set_table(unit_table(V, H).to("cuda:0"))
    batch, _ = ragged_batch(B, L, V, seed=1)
    rows = list(D.shard_rows(B, rank, world))
    shard = {k: v[rows].to("cuda:0") for k, v in batch.items()}
    # this rank's LOCAL gradient from an independent single-rank replica (same weights, same shard; the kernels are
    # deterministic), so that nothing has to look at the product's gradient between its backward and its exchange
    solo = X.RecommenderLightningModule(conf)
    solo.model = X.RecommenderModel(conf, device="cuda:0", precision="fp32", seed=3)
    solo.configure_model()
    solo.model.set_table(unit_table(V, H).to("cuda:0"))
    solo.train()
    solo.training_step(shard, 0).backward()
    solo.on_train_batch_end(None, shard, 0)
    local = solo.model.flat.grad.detach().clone()
    # the exchange in its two forms: one message behind the backward (default), or XFMR_ALLREDUCE_HALVES=1: two halves, the
    # upper layers' on a communication stream that waits for xfmr_encoder_cfg.grads_half_event only
    os.environ["XFMR_ALLREDUCE_HALVES"] = "0" if single else "1"
    trainer = X.Trainer(mod, world_size=world)
    assert trainer.optimizer.param_groups[0]["grad_scale"] == 1.0 / world
    p0 = mod.model.flat.detach().clone()
    assert (trainer.exchange is None) == single
    loss = trainer.fit_step(shard)  # no host synchronisation anywhere inside the step (the r3 spy had one: ADVICE)
    torch.cuda.synchronize()
    reduced = mod.model.flat.grad.detach().clone()  # AdamW reads the gradient, it does not change it
    torch.save({"p0": p0.cpu(), "p1": mod.model.flat.detach().cpu(), "local": local.cpu(),
                "reduced": reduced.cpu(), "rows": rows, "loss": float(loss)}, os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("train_loss,single", [("InfoNCELoss", False), ("PairwiseLogisticLoss", False), ("InfoNCELoss", True)])
def test_two_rank_fit_step_equals_averaged_independent_gradients(tmp_path, train_loss, single):
    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, str(tmp_path), train_loss, single), nprocs=world, join=True,
                       start_method="spawn")
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=True) for i in range(world)]
    assert sorted(r[0]["rows"] + r[1]["rows"]) == list(range(8)) and not set(r[0]["rows"]) & set(r[1]["rows"])
    assert torch.equal(r[0]["p0"], r[1]["p0"])  # replicas start equal
    assert not torch.allclose(r[0]["local"], r[1]["local"])  # shards (and their rank-local negatives) really differ
    total = r[0]["local"] + r[1]["local"]
    for i in range(world):
        torch.testing.assert_close(r[i]["reduced"], total, rtol=1e-6, atol=1e-7)  # ONE SUM all-reduce of the flat buffer
    # DDP semantics: AdamW on the AVERAGE of the per-rank gradients (lr 1e-3, wd 0.01: trainer.py:327-332)
    ref = torch.nn.Parameter(r[0]["p0"].clone())
    opt = torch.optim.AdamW([ref], lr=1e-3, weight_decay=0.01)
    ref.grad = total / world
    opt.step()
    for i in range(world):
        torch.testing.assert_close(r[i]["p1"], ref.detach(), rtol=1e-5, atol=2e-7)
    assert torch.equal(r[0]["p1"], r[1]["p1"])  # replicas stay equal bit for bit


@pytest.mark.parametrize("B,nL,form", [(330, 4, "side"), (330, 3, "side"), (24, 4, "inline"), (24, 1, "inline"), (330, 1, "side")])
def test_grads_half_event_releases_a_finished_upper_half(B, nL, form):
    """The only invariant HalvedAllReduce adds: when xfmr_encoder_cfg.grads_half_event fires, the tail of the flat
    gradient (layers >= L/2, from xfmr_param_half_offset on) is COMPLETE -- although the rest of the backward is still
    running. A second stream that waits for the event ONLY copies the tail; it must equal the finished buffer bit for
    bit. Both forms of the backward: weight-gradient GEMMs on the context's side stream (>= 40 960 tokens: the early
    reduction launch is enqueued there, behind that layer's GEMMs) and everything in line; an odd layer count and a
    single layer (boundary = 0: the whole buffer is the "tail", released by the backward's last reduction)."""
    import ctypes

    for p in (ROOT, ROOT / "transformer-recommenders_amd", ROOT / "tests"):
        if str(p) not in sys.path:
            sys.path.insert(0, str(p))
    from helpers import unit_table
    from xfmr_rec_amd import _native as N
    from xfmr_rec_amd import ops

    dev = "cuda"
    L, H, A, V, I = 200, 128, 4, 500, 512
    lib = N.load()
    g = torch.Generator().manual_seed(5)
    table = unit_table(V, H).to(dev)
    ev = ctypes.c_void_p()
    N.check(lib.xfmr_event_create(ctypes.byref(ev), 0), "xfmr_event_create")
    ctx = ops.Context(dev) if form == "side" else None
    kw = dict(batch=B, seq_len=L, hidden=H, heads=A, inter=I, layers=nL, max_pos=L, precision="bf16", hidden_dropout=0.1,
              attn_dropout=0.1, seed=3, context=ctx.handle if ctx else None)
    cfg = ops.make_encoder_cfg(**kw, grads_half_event=ev.value)
    cfg_plain = ops.make_encoder_cfg(**kw)
    boundary = int(lib.xfmr_param_half_offset(ctypes.byref(cfg)))
    n_params = lib.xfmr_param_count(ctypes.byref(cfg))
    assert 0 <= boundary < n_params and (boundary == 0) == (nL == 1)
    flat = (0.05 * torch.randn(n_params, generator=g)).to(dev)
    idx = torch.randint(1, V + 1, (B, L), generator=g).to(dev)
    d_out = torch.randn(B, L, H, generator=g).to(dev)
    tok, key_mask, acts = ops.encoder_fwd(cfg_plain, flat, idx, table)
    want = ops.encoder_bwd(cfg_plain, flat, d_out.clone(), key_mask, acts)
    torch.cuda.synchronize()
    comm = torch.cuda.Stream()
    for _ in range(4):
        grads = torch.full_like(flat, float("nan"))
        tail_copy = torch.empty(n_params - boundary, device=dev)
        d = d_out.clone()
        torch.cuda.synchronize()
        ops.encoder_bwd(cfg, flat, d, key_mask, acts, grads=grads)
        N.check(lib.xfmr_stream_wait_event(comm.cuda_stream, ev.value), "xfmr_stream_wait_event")
        with torch.cuda.stream(comm):
            tail_copy.copy_(grads[boundary:])  # ordered behind the event only
        torch.cuda.synchronize()
        assert torch.equal(grads, want)  # the event changes nothing about the result
        assert torch.equal(tail_copy, want[boundary:])  # ... and the tail was final when it fired
    lib.xfmr_event_destroy(ev.value)
    if ctx:
        ctx.close()


def test_allreduce_flat_through_the_c_abi_on_a_one_rank_communicator():
    """`xfmr_allreduce_flat` (include/xfmr_hip.h K19: RCCL resolved at run time): a world of ONE rank is what one GPU can
    run -- RCCL initialises, the call is enqueued on the caller's stream and a SUM over one rank returns the buffer bit for
    bit. More ranks need more devices (RCCL refuses two ranks on one): the driver's multi-GPU bench is where that runs."""
    for p in (ROOT, ROOT / "transformer-recommenders_amd"):
        sys.path.insert(0, str(p))
    from xfmr_rec_amd import distributed as D

    torch.cuda.set_device(0)
    ex = D.AbiAllReduce(torch.device("cuda:0"))
    try:
        assert ex.world == 1 and ex.comm
        g = torch.randn(819200, device="cuda:0")  # config 2's flat gradient: 3.1 MiB
        want = g.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):  # the caller's CURRENT stream is the one it is enqueued on
            out = ex.reduce_(g)
        side.synchronize()
        assert out is g and torch.equal(g, want)
        with pytest.raises(RuntimeError, match="fp32"):
            ex.reduce_(g.double())
    finally:
        ex.close()
