"""Oracle: the reference's per-row sequence sampling and collate (TEST INFRASTRUCTURE).

numpy restatement of ``SeqDataset.sample_sequence / sample_positives / sample_negatives / __getitem__ / collate``
(``xfmr_rec/data.py:669-805``; ``data.py`` itself needs polars / lightning / sentence-transformers and cannot be
imported here). The device sampler uses a different random generator, so the two are compared through the
invariants in :func:`check_example` and through frequencies, not value by value.
"""

from __future__ import annotations

import numpy as np


def sample_sequence(rng, history, max_seq_length):  # data.py:669-688
    indices = np.arange(len(history) - 1)
    if len(indices) <= max_seq_length:
        return indices
    return np.sort(rng.choice(indices, size=max_seq_length, replace=False))


def sample_positives(rng, history, label, sampled, pos_lookahead):  # data.py:690-722
    positives = np.zeros_like(sampled)
    for i, idx in enumerate(sampled):
        start = idx + 1
        end = start + pos_lookahead if pos_lookahead > 0 else None
        cand = history[start:end][label[start:end]]
        if len(cand) > 0:
            positives[i] = rng.choice(cand)
    return positives


def sample_negatives(rng, history, sampled, n_items):  # data.py:724-747
    seq_len = len(sampled)
    cand = sorted(set(range(1, n_items + 1)) - set(history.tolist()))
    if len(cand) == 0:
        cand = list(range(1, n_items + 1))
    return rng.choice(cand, seq_len, replace=len(cand) < seq_len)


def get_item(rng, history, label, *, max_seq_length, pos_lookahead, n_items):  # data.py:749-787 (index tensors)
    history, label = np.asarray(history, dtype=np.int64), np.asarray(label, dtype=bool)
    sampled = sample_sequence(rng, history, max_seq_length)
    return {
        "positions": sampled,
        "history_item_idx": history[sampled],
        "pos_item_idx": sample_positives(rng, history, label, sampled, pos_lookahead),
        "neg_item_idx": np.asarray(sample_negatives(rng, history, sampled, n_items), dtype=np.int64),
    }


def collate(examples):  # data.py:789-805: pad_sequence(batch_first=True) = right-pad with 0 to the longest row
    width = max(1, max(len(e["history_item_idx"]) for e in examples))
    out = {}
    for key in ("history_item_idx", "pos_item_idx", "neg_item_idx"):
        a = np.zeros((len(examples), width), dtype=np.int64)
        for r, e in enumerate(examples):
            a[r, : len(e[key])] = e[key]
        out[key] = a
    return out


def check_example(history, label, hist_row, pos_row, neg_row, *, max_seq_length, pos_lookahead, n_items):
    """Assert everything ``SeqDataset.__getitem__`` + ``collate`` guarantee about one (padded) row."""
    history, label = np.asarray(history, dtype=np.int64), np.asarray(label, dtype=bool)
    n = len(history)
    cnt = min(max(n - 1, 0), max_seq_length)
    assert (hist_row[cnt:] == 0).all() and (pos_row[cnt:] == 0).all() and (neg_row[cnt:] == 0).all(), "padding"
    hist, pos, neg = hist_row[:cnt], pos_row[:cnt], neg_row[:cnt]
    # the sampled history is a strictly increasing subsequence of positions 0..n-2: recover it greedily
    if n - 1 <= max_seq_length:
        positions = np.arange(cnt)
        assert (hist == history[:cnt]).all(), "short rows keep every position"
    else:
        positions, p = [], 0
        for v in hist:
            while p < n - 1 and history[p] != v:
                p += 1
            assert p < n - 1, "history item out of order / not in the row"
            positions.append(p)
            p += 1
        positions = np.asarray(positions)
    hist_set = set(history.tolist())
    for k, p in enumerate(positions):
        start = p + 1
        end = start + pos_lookahead if pos_lookahead > 0 else None
        cand = history[start:end][label[start:end]]
        if n - 1 > max_seq_length:
            # the greedy position may be earlier than the sampled one when items repeat: accept any later window
            later = history[start:][label[start:]]
            assert pos[k] == 0 or pos[k] in later
        elif len(cand) == 0:
            assert pos[k] == 0, "no positive candidate -> 0"
        else:
            assert pos[k] in cand, "positive must be a later positive-labelled item in the window"
    n_cand = n_items - len(hist_set & set(range(1, n_items + 1)))
    if n_cand > 0:
        assert not (set(neg.tolist()) & hist_set), "negatives must not be in the history"
    assert ((neg >= 1) & (neg <= n_items)).all()
    if (n_cand if n_cand > 0 else n_items) >= cnt:
        assert len(set(neg.tolist())) == cnt, "negatives are drawn without replacement"
    return positions
