#!/usr/bin/env python3
"""Throughput of the device sequence sampler (xfmr_seq_sample) beside the numpy oracle (the reference's per-row
sampling, xfmr_rec/data.py:669-805) on MovieLens-1M-shaped synthetic histories.

    python scripts/bench_sampler.py [--users 6040] [--items 3883] [--seq-len 200] [--batch 128] [--iters 200]
"""
import argparse
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "transformer-recommenders_amd"):
    sys.path.insert(0, str(p))

import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=6040)
    ap.add_argument("--items", type=int, default=3883)
    ap.add_argument("--seq-len", type=int, default=200)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--cpu-rows", type=int, default=512)
    a = ap.parse_args()
    from oracle import sampler as OS
    from xfmr_rec_amd.data import DeviceSeqDataset, SeqDataConfig

    rng = np.random.default_rng(0)
    lens = np.clip(np.round(np.exp(rng.normal(4.35, 1.0, a.users))), 20, 2000).astype(int)  # ML-1M-like
    hs = [rng.integers(1, a.items + 1, n) for n in lens]
    ls = [np.concatenate([rng.random(n - 1) < 0.58, [True]]) for n in lens]
    cfg = SeqDataConfig(max_seq_length=a.seq_len, pos_lookahead=0)
    ds = DeviceSeqDataset.from_events(cfg, hs, ls, a.items)
    rows = [rng.integers(0, len(ds), a.batch) for _ in range(a.iters)]
    for i in range(5):
        ds.sample_batch(rows[i], i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.iters):
        ds.sample_batch(rows[i], 100 + i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    gpu = a.iters * a.batch / dt
    orng = np.random.default_rng(1)
    t0 = time.perf_counter()
    ex = [OS.get_item(orng, hs[i % a.users], ls[i % a.users], max_seq_length=a.seq_len, pos_lookahead=0,
                      n_items=a.items) for i in range(a.cpu_rows)]
    OS.collate(ex[: a.batch])
    cpu = a.cpu_rows / (time.perf_counter() - t0)
    print(f"rows in dataset {len(ds)} (mean history {lens.mean():.0f}); device sampler {gpu:,.0f} sequences/s "
          f"({dt / a.iters * 1e3:.3f} ms per batch of {a.batch}, host-side call included); numpy oracle, 1 core: "
          f"{cpu:,.0f} sequences/s -> x{gpu / cpu:,.0f}")


if __name__ == "__main__":
    main()
