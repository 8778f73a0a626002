"""Device-side twin of the reference's ``SeqDataset`` sampling (``xfmr_rec/data.py:543-805``).

The reference samples one row at a time on the host (numpy) inside DataLoader workers and pads in ``collate``;
at the step rates of the MI355X path (> 6e4 sequences/s) that is the bottleneck. Here
the processed histories live in HBM as one CSR (``items``, ``labels``, ``offsets``) and a whole batch -- sampled
positions, positives, negatives, right-padded to the batch's longest row -- is produced by ONE kernel launch
(``xfmr_seq_sample``), already resident where the training step reads it.

Not carried over: the ``*_item_text`` string lists of ``SeqBatch`` (``data.py:778-785``) -- the training step never
reads them -- and the MovieLens download / polars ETL (``data.py:60-515``: out of scope).
"""

from __future__ import annotations

import numpy as np
import pydantic
import torch

from . import _native as N


class SeqDataConfig(pydantic.BaseModel):
    """``data.py:543-545``."""

    max_seq_length: int = 32
    pos_lookahead: int = 0


class DeviceSeqDataset:
    """Histories resident in HBM + batched sampling.

    ``histories[r]`` / ``labels[r]``: item indices (1..n_items, 0 is padding) and positive-label flags of row ``r``
    in time order, i.e. what ``SeqDataset.process_events`` produces (``data.py:590-656``), including its trimming of
    events after the last positive and its duplication of long rows (``duplicate_rows``) -- see :meth:`from_events`.
    """

    def __init__(self, config: SeqDataConfig, histories, labels, n_items: int, device="cuda"):
        self.config = config
        self.n_items = int(n_items)
        lens = np.asarray([len(h) for h in histories], dtype=np.int64)
        if len(lens) == 0 or lens.min() <= 0:
            raise ValueError("rows must be non-empty (data.py:654 filters empty histories)")
        self.lengths = lens
        off = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=off[1:])
        self.device = torch.device(device)
        self.items = torch.from_numpy(np.concatenate([np.asarray(h, dtype=np.int64) for h in histories])).to(self.device)
        self.labels = torch.from_numpy(
            np.concatenate([np.asarray(l, dtype=np.uint8) for l in labels])).to(self.device)
        self.offsets = torch.from_numpy(off).to(self.device)

    @classmethod
    def from_events(cls, config: SeqDataConfig, item_idx_rows, label_rows, n_items: int, device="cuda"):
        """Apply ``map_id2idx``'s trimming and ``duplicate_rows`` (``data.py:590-636``) to raw per-user rows."""
        hs, ls = [], []
        for h, l in zip(item_idx_rows, label_rows):
            h, l = np.asarray(h, dtype=np.int64), np.asarray(l, dtype=bool)
            end = (np.flatnonzero(l).max(initial=-1)) + 1  # trim events after the last positive label
            h, l = h[:end], l[:end]
            if len(h) == 0:
                continue
            copies = (len(h) - 1) // config.max_seq_length + 1
            hs += [h] * copies
            ls += [l] * copies
        return cls(config, hs, ls, n_items, device)

    def __len__(self) -> int:
        return len(self.lengths)

    def sample_batch(self, rows, seed: int) -> dict[str, torch.Tensor]:
        """``collate([dataset[r] for r in rows])`` (``data.py:749-805``) in one launch; returns the three index tensors
        of ``SeqBatch`` on the device, shape ``(len(rows), width)``."""
        rows_np = np.asarray(rows, dtype=np.int64)
        lens = self.lengths[rows_np]
        width = int(max(1, min(self.config.max_seq_length, int(lens.max()) - 1)))
        B = len(rows_np)
        rows_t = torch.from_numpy(rows_np).to(self.device)
        out = [torch.empty((B, width), dtype=torch.int64, device=self.device) for _ in range(3)]
        lib = N.load()
        nbytes = lib.xfmr_seq_sample_workspace(B, self.n_items)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        N.check(
            lib.xfmr_seq_sample(N.ptr(self.items), N.ptr(self.labels), N.ptr(self.offsets), N.ptr(rows_t), B, width,
                                self.config.max_seq_length, self.config.pos_lookahead, self.n_items, int(lens.max()),
                                int(seed) & (2**64 - 1), N.ptr(out[0]), N.ptr(out[1]), N.ptr(out[2]), N.ptr(ws), nbytes,
                                N.stream()),
            "xfmr_seq_sample",
        )
        return {"history_item_idx": out[0], "pos_item_idx": out[1], "neg_item_idx": out[2]}
