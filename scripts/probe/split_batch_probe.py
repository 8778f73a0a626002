#!/usr/bin/env python3
"""Probe: the encoder of a step run as TWO half batches on two streams (whole sequences each), so that one half's
partly filled last tile round runs beside the other half's kernels. Timing only (dropout row keys restart per half).
usage: python scripts/probe/split_batch_probe.py [bench.py args]   (XFMR_SPLIT=0: the unsplit step, same flags)"""
import os, pathlib, runpy, sys
ROOT = pathlib.Path(__file__).resolve().parents[2]
for p in (ROOT, ROOT / "transformer-recommenders_amd"):
    sys.path.insert(0, str(p))
import torch
from xfmr_rec_amd import models as M

if os.environ.get("XFMR_SPLIT", "1") != "0":
    orig = M.RecommenderModel._encode_tokens
    streams = {}

    def split(self, item_idx=None, item_embeds=None, embed_event=None, packed=None):
        if packed is not None or item_idx is None or item_idx.shape[0] < 2 or not item_idx.is_cuda:
            return orig(self, item_idx, item_embeds, embed_event, packed)
        if "s" not in streams:
            streams["s"] = (torch.cuda.Stream(), torch.cuda.Stream())
        s1, s2 = streams["s"]
        cur = torch.cuda.current_stream()
        half = item_idx.shape[0] // 2
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            t1, k1 = orig(self, item_idx[:half], None, None, None)
        with torch.cuda.stream(s2):
            t2, k2 = orig(self, item_idx[half:], None, None, None)
        cur.wait_stream(s1); cur.wait_stream(s2)
        for t in (t1, k1, t2, k2):
            t.record_stream(cur)
        return torch.cat([t1, t2], 0), torch.cat([k1, k2], 0)

    M.RecommenderModel._encode_tokens = split

sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(str(ROOT / "bench.py"), run_name="__main__")
