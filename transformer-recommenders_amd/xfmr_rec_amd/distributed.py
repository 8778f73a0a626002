"""Data-parallel glue: one process per GPU, sequences sharded by rank, ONE collective per step.

The reference reaches data parallelism implicitly (Lightning ``strategy: auto`` -> torch DDP,
``config.yaml:5-6,35``): bucketed gradient all-reduce (SUM, then / world) and a ``DistributedSampler``.
Here every trainable tensor and its gradient live in one flat buffer, so the exchange is one
``all_reduce`` over RCCL/xGMI (3.1 MiB at the MovieLens-1M config: latency-bound, buckets would only add
latencies) -- or two halves, the upper layers' first, underneath the lower layers' backward
(:class:`HalvedAllReduce`) -- and the 1/world factor is folded into the fused AdamW launch. The item table is
frozen and replicated; negatives stay rank-local exactly as under the reference's DDP.
"""

from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: str | None = None) -> tuple[int, int, int]:
    """(rank, local_rank, world). RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from torchrun."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if rehearsal_on_one_gpu():
            backend, local = "gloo", 0  # RCCL refuses two ranks on one device
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def rehearsal_on_one_gpu() -> bool:
    """XFMR_REHEARSE_ONE_GPU=1: run a multi-rank job's ranks on device 0 with the gloo backend -- exercises the
    N > 1 control flow (rendezvous, barriers, the all-reduce call, max-over-ranks timing) where only one GPU is at
    hand. Never a measurement configuration."""
    return os.environ.get("XFMR_REHEARSE_ONE_GPU", "0") == "1"


def shard_rows(n_rows: int, rank: int, world: int) -> range:
    """Rows of the epoch owned by ``rank`` (DistributedSampler without shuffling: rank::world)."""
    return range(rank, n_rows, world)


def allreduce_flat_grad_(flat_grad: torch.Tensor, group=None) -> torch.Tensor:
    """SUM all-reduce of the flat gradient in place (the only collective of a training step)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    return flat_grad


class AbiAllReduce:
    """The exchange through the C ABI (``xfmr_allreduce_flat``, ``include/xfmr_hip.h`` K19) instead of ``torch.distributed``:
    what a host that is not PyTorch binds. The RCCL communicator is created from a 128-byte id that rank 0 makes
    (``xfmr_comm_unique_id``) and the already initialised process group ships (any backend: it is 128 bytes, once);
    afterwards a step's exchange is ONE call on the compute stream -- no Python collective, no work object.
    ``XFMR_ALLREDUCE=abi`` selects it in :class:`~xfmr_rec_amd.trainer.Trainer`; the default stays ``torch.distributed``
    (the same RCCL underneath, and the driver's multi-GPU runs have only ever exercised that one)."""

    def __init__(self, device, group=None):
        import ctypes

        from . import _native as N

        self._lib, self._N = N.load(), N
        self.comm = None
        world = dist.get_world_size(group) if dist.is_initialized() else 1
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = world
        ident = (ctypes.c_ubyte * N.COMM_ID_BYTES)()
        if rank == 0:
            N.check(self._lib.xfmr_comm_unique_id(ident), "xfmr_comm_unique_id")
        if world > 1:
            box = [bytes(ident)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = (ctypes.c_ubyte * N.COMM_ID_BYTES).from_buffer_copy(box[0])
        comm = ctypes.c_void_p()
        with torch.cuda.device(device):
            N.check(self._lib.xfmr_comm_create(ctypes.byref(comm), ident, world, rank), "xfmr_comm_create")
        self.comm = comm.value

    def reduce_(self, flat_grad: torch.Tensor) -> torch.Tensor:
        N = self._N
        if not flat_grad.is_cuda or not flat_grad.is_contiguous() or flat_grad.dtype != torch.float32:
            raise RuntimeError("xfmr_allreduce_flat takes a contiguous fp32 buffer in HBM")
        N.check(self._lib.xfmr_allreduce_flat(self.comm, flat_grad.data_ptr(), flat_grad.numel(), N.stream()),
                "xfmr_allreduce_flat")
        return flat_grad

    def close(self):
        if self.comm is not None:
            self._lib.xfmr_comm_destroy(self.comm)
            self.comm = None


class HalvedAllReduce:
    """The gradient exchange of a step in TWO messages, the first one overlapped with the backward (SURVEY section 8e:
    "overlapped with the last layers' backward"; the reference gets the same from torch DDP's buckets, ``config.yaml:5-6``).

    ``xfmr_encoder_bwd`` finishes the gradients of layers >= layers / 2 -- the contiguous tail of the flat buffer from
    ``boundary`` on -- as soon as that layer's backward is enqueued and records ``event`` behind them
    (``xfmr_encoder_cfg.grads_half_event``). :meth:`reduce_` is called once the backward has been enqueued: the tail's
    all-reduce goes to a communication stream that waits for the event only -- so on the GPU it runs beside the lower
    layers' backward --, the head's follows the backward on the caller's stream, which then also waits for the tail. The
    result equals one SUM all-reduce of the whole buffer (``tests/test_host_logic.py``). On CPU tensors (gloo) the two
    slices are simply reduced one after the other."""

    def __init__(self, model, group=None):
        import ctypes

        from . import _native as N
        from . import ops

        self.group = group
        self.event = None
        self.comm = None
        c = model.config
        cfg = ops.make_encoder_cfg(batch=1, seq_len=1, hidden=c.hidden_size, heads=c.num_attention_heads,
                                   inter=c.intermediate_size, layers=c.num_hidden_layers, max_pos=c.max_seq_length,
                                   precision=model.precision)
        self.boundary = int(N.load().xfmr_param_half_offset(ctypes.byref(cfg)))
        if model.flat.is_cuda:
            e = ctypes.c_void_p()
            with torch.cuda.device(model.device):
                N.check(N.load().xfmr_event_create(ctypes.byref(e), 0), "xfmr_event_create")
                self.comm = torch.cuda.Stream(device=model.device)
            self.event = e.value
            self._lib = N.load()
        model.grads_half_event = self.event  # picked up by RecommenderModel._cfg

    def reduce_(self, flat_grad: torch.Tensor) -> torch.Tensor:
        if not (dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1):
            return flat_grad
        b = self.boundary
        head, tail = flat_grad[:b], flat_grad[b:]
        if self.event is None or not flat_grad.is_cuda:
            dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)
            if b > 0:
                dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group)
            return flat_grad
        from . import _native as N
        from . import ops

        if flat_grad.data_ptr() != ops.last_encoder_grad_ptr:
            # `grads_half_event` says when xfmr_encoder_bwd finished the upper half of ITS output buffer. That is `.grad`
            # itself only when the step started from `.grad = None` (zero_grad(set_to_none=True): autograd then adopts the
            # backward's buffer); with an accumulating `.grad`, AccumulateGrad's += runs after the whole backward and the
            # event would release half-written sums. One message behind the backward instead.
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
            return flat_grad
        main = torch.cuda.current_stream()
        N.check(self._lib.xfmr_stream_wait_event(self.comm.cuda_stream, self.event), "xfmr_stream_wait_event")
        with torch.cuda.stream(self.comm):
            dist.all_reduce(tail, op=dist.ReduceOp.SUM, group=self.group)  # beside the lower layers' backward
        if b > 0:
            dist.all_reduce(head, op=dist.ReduceOp.SUM, group=self.group)  # behind the backward's last launch
        main.wait_stream(self.comm)
        tail.record_stream(self.comm)
        return flat_grad

    def close(self):
        if self.event is not None:
            self._lib.xfmr_event_destroy(self.event)
            self.event = None
