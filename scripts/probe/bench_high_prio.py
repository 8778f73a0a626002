"""bench.py with the whole step on a HIGH-priority stream (the logging pass / weight-gradient GEMMs stay on their lowest-
priority streams): does a third priority level protect the main chain's kernels better than "normal vs lowest"?
    python scripts/probe/bench_high_prio.py [0|1] <bench args>"""
import pathlib
import runpy
import sys

import torch

ROOT = pathlib.Path(__file__).resolve().parents[2]
high = sys.argv[1] == "1"
sys.argv = [str(ROOT / "bench.py")] + sys.argv[2:]
if high:
    lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
    s = torch.cuda.Stream(priority=-1)
    with torch.cuda.stream(s):
        runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
else:
    runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
