"""Sequence sampler.

CPU: the numpy oracle (oracle/sampler.py, a restatement of xfmr_rec/data.py:669-805) satisfies the invariant checker.
GPU: the device sampler (xfmr_seq_sample) satisfies the same invariants on ragged synthetic histories, is reproducible
per seed, and its sampling frequencies -- positions, positives, negatives -- agree with the oracle's (the two use
different random generators, so parity is distributional: total-variation distance between empirical distributions).
"""

import numpy as np
import pytest
import torch

from oracle import sampler as OS

V = 57


def _rows(seed=0):
    rng = np.random.default_rng(seed)
    lens = [1, 2, 3, 9, 17, 33, 40, 64, 120, 300]
    hs, ls = [], []
    for n in lens:
        h = rng.integers(1, V + 1, n)
        l = rng.random(n) < 0.6
        l[-1] = True  # process_events trims everything after the last positive
        hs.append(h)
        ls.append(l)
    # a row whose history covers the whole catalogue (negatives then come from all items, with replacement if needed)
    hs.append(np.concatenate([np.arange(1, V + 1), rng.integers(1, V + 1, 20)]))
    ls.append(np.ones(V + 20, dtype=bool))
    return hs, ls


@pytest.mark.parametrize("lookahead", [0, 3])
def test_oracle_sampler_satisfies_the_reference_invariants(lookahead):
    hs, ls = _rows()
    rng = np.random.default_rng(1)
    for _ in range(20):
        ex = [OS.get_item(rng, h, l, max_seq_length=32, pos_lookahead=lookahead, n_items=V) for h, l in zip(hs, ls)]
        batch = OS.collate(ex)
        assert batch["history_item_idx"].shape[1] == 32
        for r, (h, l) in enumerate(zip(hs, ls)):
            OS.check_example(h, l, batch["history_item_idx"][r], batch["pos_item_idx"][r], batch["neg_item_idx"][r],
                             max_seq_length=32, pos_lookahead=lookahead, n_items=V)


def _tv(a, b):
    a, b = a / a.sum(), b / b.sum()
    return 0.5 * np.abs(a - b).sum()


@pytest.mark.gpu
@pytest.mark.parametrize("lookahead", [0, 3])
def test_device_sampler_invariants_and_reproducibility(lookahead):
    from xfmr_rec_amd.data import DeviceSeqDataset, SeqDataConfig

    hs, ls = _rows()
    ds = DeviceSeqDataset(SeqDataConfig(max_seq_length=32, pos_lookahead=lookahead), hs, ls, V)
    rows = np.arange(len(hs))
    first = None
    for seed in range(12):
        b = {k: v.cpu().numpy() for k, v in ds.sample_batch(rows, seed).items()}
        assert b["history_item_idx"].shape == (len(hs), 32)
        # the rows' lengths, known on the host before the launch (the packed layout of the training step needs them there)
        assert (b["lengths"] == (b["history_item_idx"] != 0).sum(1)).all()
        assert ((b["history_item_idx"] != 0) == (np.arange(32)[None, :] < b["lengths"][:, None])).all()  # right-padded
        for r, (h, l) in enumerate(zip(hs, ls)):
            OS.check_example(h, l, b["history_item_idx"][r], b["pos_item_idx"][r], b["neg_item_idx"][r],
                             max_seq_length=32, pos_lookahead=lookahead, n_items=V)
        if seed == 0:
            first = b
            again = {k: v.cpu().numpy() for k, v in ds.sample_batch(rows, 0).items()}
            assert all((again[k] == b[k]).all() for k in b)
        elif seed == 1:
            assert any((first[k] != b[k]).any() for k in b)
    # width follows the batch's longest row (pad_sequence), not max_seq_length
    small = ds.sample_batch(np.array([1, 2, 3]), 5)
    assert small["history_item_idx"].shape == (3, 8)


@pytest.mark.gpu
def test_device_sampler_frequencies_match_the_oracle():
    from xfmr_rec_amd.data import DeviceSeqDataset, SeqDataConfig

    rng = np.random.default_rng(7)
    n, L, look = 90, 16, 5
    h = rng.permutation(np.arange(1, n + 1)) % V + 1
    l = rng.random(n) < 0.5
    l[-1] = True
    ds = DeviceSeqDataset(SeqDataConfig(max_seq_length=L, pos_lookahead=look), [h], [l], V)
    reps = 4000
    dev_neg = np.zeros(V + 1)
    got = 0
    # 64 copies of the row: every dataset row has its own random stream
    ds64 = DeviceSeqDataset(SeqDataConfig(max_seq_length=L, pos_lookahead=look), [h] * 64, [l] * 64, V)
    for seed in range(reps // 64 + 1):
        b = {k: v.cpu().numpy() for k, v in ds64.sample_batch(np.arange(64), seed).items()}
        for r in range(64):
            OS.check_example(h, l, b["history_item_idx"][r], b["pos_item_idx"][r], b["neg_item_idx"][r],
                             max_seq_length=L, pos_lookahead=look, n_items=V)
            np.add.at(dev_neg, b["neg_item_idx"][r], 1)
        got += 64
    orng = np.random.default_rng(11)
    ora_neg = np.zeros(V + 1)
    for _ in range(got):
        e = OS.get_item(orng, h, l, max_seq_length=L, pos_lookahead=look, n_items=V)
        np.add.at(ora_neg, e["neg_item_idx"], 1)
    # negatives: uniform over the catalogue minus the history, in both
    assert _tv(dev_neg[1:] + 1e-9, ora_neg[1:] + 1e-9) < 0.03
    allowed = np.setdiff1d(np.arange(1, V + 1), h)
    if len(allowed):
        assert dev_neg[allowed].sum() == dev_neg.sum()
        f = dev_neg[allowed] / dev_neg[allowed].sum()
        assert np.abs(f - 1 / len(allowed)).max() < 0.25 / len(allowed) + 0.01


@pytest.mark.gpu
def test_device_sampler_position_and_positive_frequencies():
    """Rows with UNIQUE items make the sampled positions recoverable: every position must be chosen with probability
    L / (n - 1), and the positive of a position uniformly among its window's positive-labelled items."""
    from xfmr_rec_amd.data import DeviceSeqDataset, SeqDataConfig

    Vb, n, L, look = 400, 60, 12, 4
    rng = np.random.default_rng(3)
    h = rng.permutation(np.arange(1, Vb + 1))[:n]
    l = rng.random(n) < 0.5
    l[-1] = True
    ds = DeviceSeqDataset(SeqDataConfig(max_seq_length=L, pos_lookahead=look), [h] * 128, [l] * 128, Vb)
    where = {int(v): i for i, v in enumerate(h)}
    cnt_pos = np.zeros(n - 1)
    pair = {}
    total = 0
    for seed in range(40):
        b = {k: v.cpu().numpy() for k, v in ds.sample_batch(np.arange(128), seed).items()}
        for r in range(128):
            ps = [where[int(v)] for v in b["history_item_idx"][r]]
            assert ps == sorted(ps) and len(set(ps)) == L
            cnt_pos[ps] += 1
            for p, pv in zip(ps, b["pos_item_idx"][r]):
                pair.setdefault(p, {}).setdefault(int(pv), 0)
                pair[p][int(pv)] += 1
            total += 1
    f = cnt_pos / total
    assert np.abs(f - L / (n - 1)).max() < 0.03, (f.min(), f.max(), L / (n - 1))
    for p, d in pair.items():
        cand = h[p + 1:p + 1 + look][l[p + 1:p + 1 + look]]
        if len(cand) == 0:
            assert set(d) == {0}
        else:
            assert set(d) <= set(int(c) for c in cand)
            tot = sum(d.values())
            if tot > 300:
                assert max(abs(d.get(int(c), 0) / tot - 1 / len(cand)) for c in cand) < 0.08
