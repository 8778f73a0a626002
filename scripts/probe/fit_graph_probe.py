"""Trainer.fit(graph="on") on the default stream vs inside a side-stream context (capture protocol probe)."""
import sys, pathlib, faulthandler
faulthandler.enable()
ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "transformer-recommenders_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import torch
import xfmr_rec_amd as X
from test_gpu_graph import _setup

mode = sys.argv[1]
mod, batches = _setup(X)
tr = X.Trainer(mod)
seq = [batches[i % len(batches)] for i in range(12)]
if mode == "side":
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        out = tr.fit(seq, graph="on")
else:
    out = tr.fit(seq, graph="on")
torch.cuda.synchronize()
print(mode, "ok", out[:4], tr.graph_choice)
