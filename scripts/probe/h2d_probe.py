#!/usr/bin/env python3
"""Where does the H2D-inclusive step lose its time? Variants of the hand-over at the bench shape, one process, one box.
usage: python scripts/probe/h2d_probe.py [--batch 512] [--steps 30]"""
import argparse, ctypes, json, pathlib, sys, time
ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "transformer-recommenders_amd"):
    sys.path.insert(0, str(p))
import torch
import bench as Bn
import xfmr_rec_amd as X
from xfmr_rec_amd import _native as N
from xfmr_rec_amd.data import PinnedBatchRing

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--steps", type=int, default=30)
a = ap.parse_args()
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
B, L, H, V = a.batch, 200, 128, 3883
conf = X.LightningConfig(hidden_size=H, num_attention_heads=4, intermediate_size=512, num_hidden_layers=4, max_seq_length=L)
mod = X.RecommenderLightningModule(conf); mod.configure_model(); mod.model.set_table(Bn.unit_table(V, H).to(dev))
tr = X.Trainer(mod); mod.train()
KEYS = ("history_item_idx", "pos_item_idx", "neg_item_idx")
host, res = [], []
for i in range(4):
    b, _ = Bn.synth_batch(B, L, V, 1000 + i, "dense")
    host.append(torch.stack([b[k] for k in KEYS]).pin_memory()); res.append({k: v.to(dev) for k, v in b.items()})
overlap = B * L >= 102400
lib = N.load()

def run_step(batch):
    tr.optimizer.zero_grad(set_to_none=True)
    out = mod.compute_losses(batch, sync_metrics=False, defer_logging=overlap)
    out["loss/InfoNCELoss"].backward()
    tr.optimizer.step(); mod.sync_logging()

def timed(fn, n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n): fn(i)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

for i in range(300): run_step(res[i % 4])
out = {}
out["resident"] = timed(lambda i: run_step(res[i % 4]), a.steps)

def make_ring(low=False):
    r = PinnedBatchRing(dev, B, L, slots=3)
    if low:
        h = ctypes.c_void_p(); N.check(lib.xfmr_low_priority_stream_create(ctypes.byref(h)), "s"); r.copy_stream = h.value
    return r

host_t = {}
def ring_variant(name, low=False, nofree=False, late=False, torch_stream=False):
    r = make_ring(low)
    if torch_stream:
        ts = torch.cuda.Stream(device=dev); r.copy_stream = ts.cuda_stream; r._ts = ts
    if nofree:
        r.used = [False] * r.slots
        orig_release = r.release
        def rel():
            orig_release(); r.used = [False] * r.slots
        r.release = rel
    acc = [0.0, 0.0]
    def fn(i):
        if r.pending == 0: r.stage(host[i % 4])
        t0 = time.perf_counter(); batch = r.take(); t1 = time.perf_counter()
        if not late: r.stage(host[(i + 1) % 4])
        t2 = time.perf_counter()
        run_step(batch)
        if late: r.stage(host[(i + 1) % 4])
        acc[0] += t1 - t0; acc[1] += t2 - t1
    for i in range(5): fn(i)
    acc[0] = acc[1] = 0.0
    out[name] = timed(fn, a.steps)
    host_t[name] = [round(x / a.steps * 1e6, 1) for x in acc]

ring_variant("ring")
ring_variant("ring_lowprio", low=True)
ring_variant("ring_nofree", nofree=True)
ring_variant("ring_late", late=True)
ring_variant("ring_torchstream", torch_stream=True)

def inline(i):
    run_step({k: host[i % 4][j].to(dev, non_blocking=True) for j, k in enumerate(KEYS)})
out["inline3"] = timed(inline, a.steps)
slot = torch.zeros((3, B, L), dtype=torch.int64, device=dev)
def inline1(i):
    slot.copy_(host[i % 4], non_blocking=True)
    run_step({k: slot[j] for j, k in enumerate(KEYS)})
out["inline1"] = timed(inline1, a.steps)
# zero-copy: the step's kernels read the page-locked host block directly? (index tensors are read by the gather and the loss prepare)
out["resident_again"] = timed(lambda i: run_step(res[i % 4]), a.steps)
print(json.dumps({"ms": {k: round(v, 4) for k, v in out.items()}, "host_us_take_stage": host_t}))
