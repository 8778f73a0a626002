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
    os.environ["XFMR_ALLREDUCE_SINGLE"] = "1" if single else "0"
    trainer = X.Trainer(mod, world_size=world)
    assert trainer.optimizer.param_groups[0]["grad_scale"] == 1.0 / world
    p0 = mod.model.flat.detach().clone()
    seen = {}
    real = trainer.allreduce_
    # the exchange in its two forms: two halves, the upper layers' overlapped with the backward (default), or one message
    assert (trainer.exchange is None) == single

    def spy(flat_grad):  # Trainer.fit_step calls its allreduce_ between backward and optimizer.step
        torch.cuda.synchronize()
        seen["local"] = flat_grad.detach().clone()
        real(flat_grad)
        torch.cuda.synchronize()
        seen["reduced"] = flat_grad.detach().clone()

    trainer.allreduce_ = spy
    loss = trainer.fit_step(shard)
    torch.cuda.synchronize()
    torch.save({"p0": p0.cpu(), "p1": mod.model.flat.detach().cpu(), "local": seen["local"].cpu(),
                "reduced": seen["reduced"].cpu(), "rows": rows, "loss": float(loss)}, os.path.join(out_dir, f"r{rank}.pt"))
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
