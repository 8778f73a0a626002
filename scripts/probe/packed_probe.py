"""Packed vs padded step on MovieLens-like ragged batches: GPU time per step (events) and host enqueue time."""
import os, sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "transformer-recommenders_amd"):
    sys.path.insert(0, str(p))
import torch
import bench
import xfmr_rec_amd as X

B, L, V, H = 512, 200, 3883, 128
dev = "cuda"
conf = X.LightningConfig(hidden_size=H, num_attention_heads=4, intermediate_size=512, num_hidden_layers=4, max_seq_length=L)
mod = X.RecommenderLightningModule(conf)
mod.configure_model()
mod.model.set_table(bench.unit_table(V, H).to(dev))
tr = X.Trainer(mod)
batches = []
for i in range(4):
    b, lens = bench.synth_batch(B, L, V, 5000 + i, "ml")
    d = {k: v.to(dev) for k, v in b.items()}
    off = torch.zeros(B + 1, dtype=torch.int64); off[1:] = torch.cumsum(torch.tensor(lens), 0)
    batches.append((d, torch.tensor(lens), off.to(dev), int(off[-1])))
for mode in ("padded", "packed", "padded", "packed"):
    def step(i):
        d, lens, off, rows = batches[i % 4]
        b = dict(d)
        if mode == "packed":
            b |= {"lengths": lens, "offsets": off, "packed_rows": rows}
        return tr.fit_step(b)
    for i in range(20): step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); host = 0.0
    for i in range(40):
        h0 = time.perf_counter(); step(i); host += time.perf_counter() - h0
    torch.cuda.synchronize()
    print(mode, "ms/step", round((time.perf_counter() - t0) / 40 * 1e3, 3), "host enqueue ms/step", round(host / 40 * 1e3, 3), flush=True)
