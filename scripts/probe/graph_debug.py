#!/usr/bin/env python3
"""Where does a hipGraph replay of the step differ from the eager step? (diagnostic)"""
import pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "transformer-recommenders_amd", ROOT / "tests"):
    sys.path.insert(0, str(p))
import torch
import xfmr_rec_amd as X
from test_gpu_graph import _setup

for dropout in (False, True):
    eager, batches = _setup(X)
    eager.model.use_device_step(True)
    tr_e = X.Trainer(eager); tr_e.optimizer.step_device = eager.model.step_device
    graphed, _ = _setup(X)
    tr_g = X.Trainer(graphed)
    if not dropout:
        import xfmr_rec_amd.models as M
        M.HIDDEN_DROPOUT_PROB = 0.0; M.ATTENTION_PROBS_DROPOUT_PROB = 0.0
    step = X.GraphedStep(tr_g, batches[0], warmup=3)
    for _ in range(3):
        tr_e.fit_step(batches[0])
    torch.cuda.synchronize()
    print("dropout", dropout, "after warmup equal:", torch.equal(eager.model.flat, graphed.model.flat),
          int(eager.model.step_device), int(graphed.model.step_device))
    for i, b in enumerate(batches[1:4]):
        le = tr_e.fit_step(b).clone()
        lg = step(b).clone()
        torch.cuda.synchronize()
        ge, gg = eager.model.flat.grad, graphed.model.flat.grad
        print(i, "loss", float(le), float(lg), "grad equal", torch.equal(ge, gg), "maxdiff", float((ge - gg).abs().max()),
              "param equal", torch.equal(eager.model.flat, graphed.model.flat), float((eager.model.flat - graphed.model.flat).abs().max()),
              "steps", int(eager.model.step_device), int(graphed.model.step_device))
        me, mg = tr_e.optimizer.state[eager.model.flat], tr_g.optimizer.state[graphed.model.flat]
        print("   exp_avg equal", torch.equal(me["exp_avg"], mg["exp_avg"]), "logged keys", len(step.logged))
