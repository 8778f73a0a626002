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
        of ``SeqBatch`` on the device, shape ``(len(rows), width)`` -- and the rows' lengths on the HOST (a history of n
        events gives min(n - 1, max_seq_length) sampled positions, ``data.py:669-688``: known before the launch), with which
        the training step runs the packed layout."""
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
        return {"history_item_idx": out[0], "pos_item_idx": out[1], "neg_item_idx": out[2],
                "lengths": torch.from_numpy(np.minimum(lens - 1, width).astype(np.int64))}


SEQ_BATCH_KEYS = ("history_item_idx", "pos_item_idx", "neg_item_idx")  # the index tensors of SeqBatch (data.py:534-540)


def row_lengths(hist: torch.Tensor) -> torch.Tensor:
    """Lengths of the right-padded rows of a (B, L) int64 CPU index tensor: L minus the row's trailing zeros (an all-padding
    row: 0). numpy on the host: ~30 us for 512 x 200."""
    import numpy as np

    a = hist.numpy()
    nz = a != 0
    last = nz.shape[1] - np.argmax(nz[:, ::-1], axis=1)
    return torch.from_numpy(np.where(nz.any(axis=1), last, 0).astype(np.int64))


class PinnedBatchRing:
    """Host -> HBM hand-over of collated batches: the step-side half of the reference's ``DataLoader(pin_memory=True)``
    + Lightning batch transfer (``xfmr_rec/data.py:915-927``).

    ``slots`` persistent device blocks of ``(3, batch, width)`` int64 (+ as many page-locked host blocks for batches
    that arrive in pageable memory), one copy stream, and per slot one "ready" and one "free" event -- all created
    ONCE. Per batch: one ``hipMemcpyAsync`` on the copy stream, two event records and one stream wait
    (``xfmr_batch_upload``); no allocation, no new event or stream object. The copy of batch i + 1 runs underneath step
    i (``stage`` is called right after ``take``). A slot is reused ``slots`` batches later; ``stage`` waits on the host
    for the step that last read it, i.e. only when the host has run a whole ring ahead of the GPU (bounded run-ahead --
    a stream-side wait in front of the copy would make the copy call itself block: ``include/xfmr_hip.h``).

        ring = PinnedBatchRing(device, batch=512, width=200)
        ring.stage(first_batch)
        for nxt in batches:              # dicts of (B, L) int64 CPU tensors, or one pinned (3, B, L) block
            dev_batch = ring.take()      # current stream waits for the copy; returns views of the device slot
            ring.stage(nxt)              # in flight while this step computes
            step(dev_batch)
    """

    def __init__(self, device, batch: int, width: int, slots: int = 6):
        import ctypes

        self.device = torch.device(device)
        self.slots = int(slots)
        if self.slots < 2:
            # with one slot the next stage() would overwrite the batch the running step is still reading (take() hands the
            # slot back only at the NEXT take / release)
            raise ValueError(f"PinnedBatchRing needs at least 2 slots, got {slots}")
        self.shape = (3, int(batch), int(width))
        self.lib = N.load()
        with torch.cuda.device(self.device):
            self.dev = [torch.zeros(self.shape, dtype=torch.int64, device=self.device) for _ in range(self.slots)]
            self.host = [torch.zeros(self.shape, dtype=torch.int64).pin_memory() for _ in range(self.slots)]
            # the rows' cumulative lengths (batch + 1 int64): what the PACKED layout of the encoder needs from the host
            # (xfmr_encoder_cfg.seq_offsets); a second small copy on the same stream, in front of the batch's
            # ... and the ORDER the packed layout takes the rows in (longest first: ops.length_order), behind the offsets:
            # [0, b] cumulative lengths in that order, [b + 1, 2 b] the order (b = the staged batch's rows)
            self.dev_off = [torch.zeros(2 * int(batch) + 1, dtype=torch.int64, device=self.device) for _ in range(self.slots)]
            self.host_off = [torch.zeros(2 * int(batch) + 1, dtype=torch.int64).pin_memory() for _ in range(self.slots)]
            h = ctypes.c_void_p()
            N.check(self.lib.xfmr_stream_create(ctypes.byref(h)), "xfmr_stream_create")
            self.copy_stream = h.value
            self.ready, self.free = [], []
            for _ in range(self.slots):
                for lst in (self.ready, self.free):
                    e = ctypes.c_void_p()
                    N.check(self.lib.xfmr_event_create(ctypes.byref(e), 0), "xfmr_event_create")
                    lst.append(e.value)
        self.used = [False] * self.slots  # slot read by a step at least once (its free event has been recorded)
        self.staged = [False] * self.slots  # slot has had a copy issued into it (its ready event has been recorded)
        self.keep = [None] * self.slots  # the copy's SOURCE block, referenced until the slot's next copy may start
        self.shapes = [self.shape] * self.slots
        self.lengths = [None] * self.slots  # per slot: the staged batch's row lengths (CPU int64) and their sum
        self.rows = [0] * self.slots
        self.head = 0  # next slot to stage into
        self.tail = 0  # next slot to take
        self.pending = 0

    def stage(self, batch, lengths=None) -> None:
        """Start the copy of one collated batch into the next slot. ``batch``: a dict with the three ``(B, L)`` int64 CPU
        index tensors (B <= batch, L <= width of this ring) or one contiguous ``(3, B, L)`` int64 CPU tensor (the collate
        output kept as one block). Page-locked input is copied from where it lies; pageable input goes through the
        slot's own page-locked block first (a host memcpy). ``lengths`` (or ``batch["lengths"]``): the rows' lengths as the
        collate knows them (it padded them, data.py:799-805); computed here from the history's trailing zeros otherwise.
        They travel to the device as cumulative offsets and come back from :meth:`take` -- the training step runs the
        packed layout with them."""
        if self.pending >= self.slots:
            raise RuntimeError("PinnedBatchRing: every slot holds a batch that has not been taken")
        s = self.head
        blk = batch if isinstance(batch, torch.Tensor) else None
        if blk is None:
            parts = [batch[k] for k in SEQ_BATCH_KEYS]
            shape = (3, *parts[0].shape)
        else:
            shape = tuple(blk.shape)
        if len(shape) != 3 or shape[0] != 3 or shape[1] > self.shape[1] or shape[2] > self.shape[2]:
            raise ValueError(f"batch of shape {shape} does not fit the ring's slots {self.shape}")
        if self.staged[s]:
            # hipMemcpyAsync from page-locked memory reads its source until the copy COMPLETES, not until the call returns:
            # the slot's previous copy (slots batches ago) must be done before its host block is rewritten or its
            # caller-supplied source released. Returns at once unless the copy stream is a whole ring behind.
            N.check(self.lib.xfmr_event_synchronize(self.ready[s]), "xfmr_event_synchronize")
            self.keep[s] = None
        if blk is None or not (blk.is_pinned() and blk.is_contiguous() and blk.dtype == torch.int64):
            dst = self.host[s].view(-1)[: shape[0] * shape[1] * shape[2]].view(shape)
            if blk is None:
                for i, p in enumerate(parts):
                    dst[i].copy_(p)
            else:
                dst.copy_(blk)
            blk = dst
        if lengths is None and not isinstance(batch, torch.Tensor):
            lengths = batch.get("lengths")
        lens = row_lengths(blk[0]) if lengths is None else torch.as_tensor(lengths, dtype=torch.int64).cpu()
        b = shape[1]
        off = self.host_off[s]
        from .ops import length_order

        order, cum = length_order(lens)  # longest first (XFMR_PACK_ORDER=0: the batch's own order)
        off[: b + 1] = cum
        off[b + 1 : 2 * b + 1] = order
        self.lengths[s], self.rows[s] = lens, int(off[b])
        nbytes = 8 * shape[0] * shape[1] * shape[2]
        N.check(self.lib.xfmr_batch_upload(self.dev_off[s].data_ptr(), off.data_ptr(), 8 * (2 * b + 1), self.copy_stream,
                                           self.free[s] if self.used[s] else None, self.ready[s]), "xfmr_batch_upload")
        N.check(self.lib.xfmr_batch_upload(self.dev[s].data_ptr(), blk.data_ptr(), nbytes, self.copy_stream, None,
                                           self.ready[s]), "xfmr_batch_upload")
        self.keep[s] = blk  # the source stays referenced until this slot's copy has completed (checked at its next stage)
        self.staged[s] = True
        self.shapes[s] = shape
        self.head = (s + 1) % self.slots
        self.pending += 1

    def take(self) -> dict[str, torch.Tensor]:
        """The oldest staged batch as device tensors (views of its slot); the current stream waits for its copy. The
        slot is handed back by the NEXT ``take`` / ``release`` (its free event is recorded on the current stream then),
        i.e. the returned tensors are valid for the step that follows this call."""
        if self.pending == 0:
            raise RuntimeError("PinnedBatchRing.take: nothing staged")
        self.release()
        s = self.tail
        N.check(self.lib.xfmr_stream_wait_event(N.stream(), self.ready[s]), "xfmr_stream_wait_event")
        self.tail = (s + 1) % self.slots
        self.pending -= 1
        self._held = s
        shape = self.shapes[s]
        v = self.dev[s].view(-1)[: shape[0] * shape[1] * shape[2]].view(shape)
        out = {k: v[i] for i, k in enumerate(SEQ_BATCH_KEYS)}
        # what the packed layout needs: the lengths (host), their cumulative offsets (device) and their sum (host)
        # (offsets = the cumulative lengths in the ORDER the packed layout takes the rows in: longest first)
        b = shape[1]
        out |= {"lengths": self.lengths[s], "offsets": self.dev_off[s][: b + 1], "order": self.dev_off[s][b + 1 : 2 * b + 1],
                "packed_rows": self.rows[s]}
        return out

    def release(self) -> None:
        """Mark the slot of the last ``take`` as read: everything enqueued on the current stream so far is ahead of the
        next copy into it."""
        s = getattr(self, "_held", None)
        if s is not None:
            N.check(self.lib.xfmr_event_record(self.free[s], N.stream()), "xfmr_event_record")
            self.used[s] = True
            self._held = None

    def close(self) -> None:
        if getattr(self, "copy_stream", None):
            torch.cuda.synchronize(self.device)
            for e in self.ready + self.free:
                self.lib.xfmr_event_destroy(e)
            self.lib.xfmr_stream_destroy(self.copy_stream)
            self.copy_stream = None
            self.ready, self.free = [], []

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
