import sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "transformer-recommenders_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import numpy as np, torch
import xfmr_rec_amd as X
from xfmr_rec_amd.data import DeviceSeqDataset, SeqDataConfig
from test_gpu_e2e import _planted_catalogue, _histories

import os
V, H, L = 400, 64, 32
B = int(os.environ.get("DBG_B", "128")); DROP = float(os.environ.get("DBG_DROP", "0.1"))
print("B", B, "dropout", DROP, "packed", os.environ.get("XFMR_PACKED", "1"))
rng = np.random.default_rng(0)
table = _planted_catalogue(V, H, seed=1)
train = _histories(V, 3000, rng)
conf = X.LightningConfig(hidden_size=H, num_attention_heads=2, intermediate_size=128, num_hidden_layers=2, max_seq_length=L,
                         train_loss="InfoNCELoss", precision="bf16", top_k=20, pooling_mode="lasttoken")
import xfmr_rec_amd.models as M
M.HIDDEN_DROPOUT_PROB = DROP; M.ATTENTION_PROBS_DROPOUT_PROB = DROP
mod = X.RecommenderLightningModule(conf); mod.configure_model(); mod.model.set_table(table.cuda())
ds = DeviceSeqDataset(SeqDataConfig(max_seq_length=L, pos_lookahead=0), train, [np.ones(len(h), bool) for h in train], n_items=V, device="cuda")
tr = X.Trainer(mod)
for step in range(1):
    rows = rng.integers(0, len(ds), size=B)
    batch = ds.sample_batch(rows, seed=step)
    h = batch["history_item_idx"].cpu()
    real = (h != 0).sum(1)
    print("step", step, "width", h.shape[1], "lens ok", bool((real == batch["lengths"]).all()), "min/max len", int(real.min()), int(real.max()),
          "right padded", bool(((h != 0) == (torch.arange(h.shape[1])[None] < real[:, None])).all()))
    mod.train(); tr.optimizer.zero_grad(set_to_none=True)
    loss = mod.training_step(batch, 0); mod.backward(loss)
    g = mod.model.flat.grad
    names = mod.model.grad_state_dict()
    bad = [k for k, v in names.items() if not torch.isfinite(v).all()]
    print("  loss", float(loss.detach()), "grad finite", bool(torch.isfinite(g).all()), "n bad", len(bad), "of", len(names), "good:", [k for k in names if k not in bad][:8])
    tr.optimizer.step(); mod.on_train_batch_end(loss, batch, 0)
    print("  params finite", bool(torch.isfinite(mod.model.flat).all()))
