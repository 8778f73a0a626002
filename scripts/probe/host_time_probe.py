#!/usr/bin/env python3
"""How far ahead of the GPU does the host run? Host time to ENQUEUE a training step (no synchronisation inside the loop)
against the step's GPU time: python scripts/probe/host_time_probe.py  (uses bench.py's own step through its module API)."""
import os
import runpy
import sys
import time

import torch

sys.argv = ["bench.py", "--steps", "40", "--warmup", "10", "--no-cpu-baseline", "--spinup-steps", "50"]
os.environ["XFMR_HOST_PROBE"] = "1"
t = time.perf_counter
import pathlib  # noqa: E402

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[2]))
import bench  # noqa: E402

orig_sync = torch.cuda.synchronize
marks = []


def sync_spy(*a, **k):
    marks.append(("sync_enter", t()))
    r = orig_sync(*a, **k)
    marks.append(("sync_exit", t()))
    return r


torch.cuda.synchronize = sync_spy
bench.main() if hasattr(bench, "main") else runpy.run_path("bench.py", run_name="__main__")
# the timed region is bracketed by synchronize calls: host enqueue time = sync_enter(after loop) - sync_exit(before loop)
pairs = [(marks[i][1], marks[i + 1][1]) for i in range(len(marks) - 1) if marks[i][0] == "sync_exit" and marks[i + 1][0] == "sync_enter"]
for a, b in pairs:
    print(f"host enqueue span between two synchronisations: {(b - a) * 1e3:8.2f} ms")
