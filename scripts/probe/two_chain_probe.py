#!/usr/bin/env python3
"""Would two interleaved half-batch chains use the GPU better than one full-batch chain? Two independent trainers of batch
256 stepped alternately on two streams (their kernels interleave on the device) against one trainer of batch 512.
(An upper-bound experiment for splitting the encoder of one step into two half-batch chains.)"""
import pathlib
import sys
import time

import torch

ROOT = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "transformer-recommenders_amd"))
import bench  # noqa: E402
import xfmr_rec_amd as X  # noqa: E402

dev = torch.device("cuda", 0)
L, H, V = 200, 128, 3883


def make(B, seed):
    conf = X.LightningConfig(hidden_size=H, num_attention_heads=4, intermediate_size=512, num_hidden_layers=4,
                             max_seq_length=L, train_loss="InfoNCELoss", precision="bf16")
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    mod.model.set_table(bench.unit_table(V, H).to(dev))
    tr = X.Trainer(mod)
    mod.train()
    b, _ = bench.synth_batch(B, L, V, seed, "dense")
    return mod, tr, {k: v.to(dev) for k, v in b.items()}


def step(mod, tr, batch, overlap):
    tr.optimizer.zero_grad(set_to_none=True)
    out = mod.compute_losses(batch, sync_metrics=False, defer_logging=overlap)
    out[f"loss/{mod.config.train_loss}"].backward()
    tr.optimizer.step()
    mod.sync_logging()


def run(label, units, n=40, overlap=True):
    streams = [torch.cuda.Stream(device=dev) for _ in units]
    for _ in range(60):
        for (mod, tr, b), s in zip(units, streams):
            with torch.cuda.stream(s):
                step(mod, tr, b, overlap)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        for (mod, tr, b), s in zip(units, streams):
            with torch.cuda.stream(s):
                step(mod, tr, b, overlap)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    seqs = n * sum(b["history_item_idx"].shape[0] for _, _, b in units)
    print(f"{label:46s} {seqs / dt:10.0f} sequences/s  {dt / n * 1e3:7.3f} ms per round")


run("one chain, batch 512", [make(512, 1)])
run("two chains, batch 256 each, two streams", [make(256, 2), make(256, 3)])
run("two chains, batch 256 each, logging in line", [make(256, 2), make(256, 3)], overlap=False)
run("one chain, batch 256", [make(256, 4)])
