"""Training-step driver: host-side mirror of ``xfmr_rec/trainer.py`` (the LightningModule surface).

``RecommenderLightningModule`` keeps the reference's method names and semantics for the training path --
``configure_model``, ``forward``, ``compute_losses`` (all seven heads + batch / logits statistics every
step, ``trainer.py:213-264``), ``training_step`` (returns ``loss/{train_loss}``, ``trainer.py:288-291``),
``configure_optimizers`` (AdamW lr 1e-3 / wd 0.01 over all parameters, ``trainer.py:327-332``) and
``state_dict`` without the frozen table (``trainer.py:352-362``). When ``lightning`` is importable it
subclasses ``lightning.pytorch.LightningModule`` so ``LightningCLI(lightning_module_cls=...)``
(``trainer.py:377-395``) accepts it; otherwise it is a plain ``nn.Module`` driven by :class:`Trainer`.

One step on the device is: gather+mask+BertEmbeddings -> N x BertLayer -> fused loss (7 heads + stats +
dL/dtok in one pass) -> encoder backward -> [flat-gradient all-reduce under DDP] -> fused AdamW. All of it is
HIP kernels from ``libxfmr_hip.so``; torch supplies memory, streams, autograd bookkeeping and RCCL.
"""

from __future__ import annotations

import contextlib
import time

import torch

from . import _native as N
from . import ops
from .losses import LOSS_CLASSES, LossConfig, LossType, stats_to_dict
from .models import ModelConfig, RecommenderModel
from .params import TOP_K

try:  # pragma: no cover - lightning is not installed in the build image
    import lightning.pytorch as _lp

    _Base = _lp.LightningModule
except Exception:  # noqa: BLE001
    _lp = None
    _Base = torch.nn.Module


class LightningConfig(LossConfig, ModelConfig):
    """``trainer.py:98-115`` minus the LanceDB index configs (retrieval is out of the hot path)."""

    train_loss: LossType = "InfoNCELoss"
    learning_rate: float = 0.001
    weight_decay: float = 0.01
    top_k: int = TOP_K
    # trainer.py:103-115: the LanceDB index configs. Accepted so that the reference's config files load unchanged; the
    # exact top-k search of this build (retrieval.py) has no index to configure.
    items_config: dict = {}
    users_config: dict = {}
    # build-specific knobs (not in the reference)
    precision: str = "bf16"  # MFMA arithmetic: "bf16" (reference default bf16-mixed) or "fp32"
    log_all_losses: bool = True  # evaluate all 7 heads + statistics every step like trainer.py:250-264
    # "in_batch": [positive | the batch's sampled negatives] (models.py:366-419, the reference's only mode);
    # "catalogue": every row of the item table is a column and pos_item_idx is the target -- the reference API's
    # EmbedLoss.forward(q, table[None].expand(N,-1,-1), target=pos_idx) with target_position=None (SURVEY F9;
    # BASELINE config 4 "full-catalogue softmax"). The (Np x V) logits are never materialised.
    negatives: str = "in_batch"


class FusedAdamW(torch.optim.Optimizer):
    """``torch.optim.AdamW`` semantics (decoupled decay, bias correction, eps outside sqrt) as one HIP launch
    over the flat parameter buffer. Parameters without a gradient are skipped, as torch does.

    ``step_device``: an int32 device tensor holding the number of completed steps. The launch then reads the step count
    from it (``xfmr_adamw_dev``) and advances it afterwards (``xfmr_step_advance``) -- no host-computed argument changes
    from step to step, so the step can be captured into a hipGraph and replayed."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0, step_device=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, grad_scale=grad_scale))
        self.step_device = step_device

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p)
                    st["exp_avg_sq"] = torch.zeros_like(p)
                st["step"] += 1
                ops.adamw_(p.data, p.grad.contiguous(), st["exp_avg"], st["exp_avg_sq"], lr=group["lr"], beta1=b1,
                           beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"], step=st["step"],
                           grad_scale=group["grad_scale"], step_device=self.step_device)
        if self.step_device is not None:
            ops.step_advance_(self.step_device)
        return loss

    def sync_step_from_device(self) -> None:
        """Write the device-side step counter back into ``state[p]["step"]``. Replays of a captured step advance the counter
        on the device only: without this a ``state_dict()`` taken after ``fit(graph="on")`` would carry a stale step, and a
        resume (or a later ``fit`` seeding its counter from it) the wrong bias corrections. One host sync."""
        if self.step_device is None:
            return
        n = int(self.step_device.item())
        for st in self.state.values():
            if "step" in st:
                st["step"] = n

    def state_dict(self):
        self.sync_step_from_device()
        return super().state_dict()


class RecommenderLightningModule(_Base):
    def __init__(self, config: LightningConfig) -> None:
        super().__init__()
        self.config = LightningConfig.model_validate(config)
        if _lp is not None:  # pragma: no cover
            self.save_hyperparameters(self.config.model_dump())
            self.strict_loading = False
        self.model: RecommenderModel | None = None
        self.items_dataset = None
        self.loss_fns: torch.nn.ModuleList | None = None
        self.logged: dict = {}

    # lightning supplies .device; the plain-module build derives it from the parameters
    def _device(self):
        if self.model is not None:
            return self.model.device
        return torch.device("cuda" if torch.cuda.is_available() else "cpu")

    def configure_model(self) -> None:
        """``trainer.py:139-161``."""
        if self.model is None:
            self.model = RecommenderModel(self.config, device=self._device(), precision=self.config.precision)
        if self.items_dataset is None and _lp is not None:  # pragma: no cover
            try:
                self.items_dataset = self.trainer.datamodule.items_dataset
            except RuntimeError:
                pass
        if self.items_dataset is not None:
            self.model.configure_embeddings(self.items_dataset)
        if self.loss_fns is None:
            self.loss_fns = self.get_loss_fns()

    def get_loss_fns(self) -> torch.nn.ModuleList:
        """``trainer.py:163-170``."""
        return torch.nn.ModuleList([cls(self.config, precision=self.config.precision) for cls in LOSS_CLASSES])

    def forward(self, item_idx: torch.Tensor) -> dict[str, torch.Tensor]:
        assert self.model is not None
        return self.model(item_idx.to(self.model.device))

    def _side_stream(self):
        if getattr(self, "_side", None) is None:
            # lowest priority: the logging pass fills what the training chain leaves idle instead of competing with it
            # (XFMR_LOG_STREAM_PRIORITY overrides; torch: lower number = higher priority, 0 = the default stream's)
            # (torch offers only "normal" and "high": the stream comes from the library. XFMR_LOG_STREAM_PRIORITY=0:
            # a normal-priority torch stream.)
            import ctypes
            import os

            if os.environ.get("XFMR_LOG_STREAM_PRIORITY", "low") == "0":
                self._side = torch.cuda.Stream(device=self.model.device)
            else:
                handle = ctypes.c_void_p()
                with torch.cuda.device(self.model.device):
                    N.check(N.load().xfmr_low_priority_stream_create(ctypes.byref(handle)), "xfmr_low_priority_stream_create")
                self._side = torch.cuda.ExternalStream(handle.value, device=self.model.device)
        return self._side

    def sync_logging(self) -> None:
        """Join the side stream of a deferred logging pass (see compute_losses(defer_logging=True))."""
        if getattr(self, "_logging_pending", False):
            torch.cuda.current_stream().wait_stream(self._side)
            self._logging_pending = False

    def compute_losses(self, batch, *, sync_metrics: bool = True, defer_logging: bool = False, profile_grad=None,
                       profile_log=None) -> dict:
        """``trainer.py:213-264``: every head's summed loss and ``...Mean``, batch and logits statistics.

        One encoder forward + ONE fused loss launch sequence produce all of it (the reference recomputes the
        logits eight times). ``loss/{train_loss}`` carries the gradient; the other heads are logging values.
        ``sync_metrics=False`` keeps the statistics on the device (no host sync in the step).
        ``defer_logging=True`` (implies device-side metrics) runs the logging heads on a side stream; the non-train
        entries of the returned dict must then only be read after ``sync_logging()``.
        ``profile_grad`` / ``profile_log``: (start, stop) hipEvent_t handles recorded around the gradient-pass /
        logging-pass kernel of the loss (``bench.py``'s roofline figures).
        """
        assert self.model is not None
        m, c = self.model, self.config
        if c.negatives not in ("in_batch", "catalogue"):
            raise ValueError(f"invalid {c.negatives = }")
        catalogue = c.negatives == "catalogue"
        if not catalogue and c.target_position != "first":
            raise ValueError("the training path scores [positive | shared negatives]: target_position='first'")
        if catalogue and c.target_position is not None:
            raise ValueError("negatives='catalogue' names the positive by `target` (= pos_item_idx): "
                             "target_position=None (losses.py:233-238)")
        dev = m.device
        hist = batch["history_item_idx"]
        L = min(hist.shape[1], m.max_seq_length)  # what _encode_tokens keeps of the history
        pos = batch["pos_item_idx"][:, -L:].to(dev, torch.int64).contiguous()
        neg = None if catalogue else batch["neg_item_idx"][:, -L:].to(dev, torch.int64).contiguous()
        # PACKED rows (xfmr_encoder_cfg.seq_offsets): when the batch carries its rows' lengths on the HOST -- the collate that
        # right-padded them knows them (data.py:799-805); PinnedBatchRing hands them over -- the encoder and the loss run on
        # each sequence's own rows only. The reference's padded layout computes the padding rows too and then drops them
        # (models.py:392); on MovieLens-like lengths that is half of the encoder's work.
        packed = None
        padded_positions = 0
        lens = batch.get("lengths")
        if torch.is_tensor(lens) and lens.is_cuda:
            lens = None  # (the launch sizes need the row count on the HOST: device-side lengths would cost a sync per step)
        if (lens is not None and hist.shape[1] == L and m.supports_packed_rows(L) and not torch.compiler.is_compiling()
                and not torch.cuda.is_current_stream_capturing()):
            rows = int(batch["packed_rows"]) if "packed_rows" in batch else int(lens.sum())
            if 0 < rows <= 0.97 * hist.shape[0] * L:  # (a nearly full batch gains nothing)
                # the sequences go into the packed layout by length, LONGEST first (ops.length_order: nothing depends on the order
                # of a batch's rows -- every loss is a sum over them -- and the attention kernels start their longest workgroups
                # first); PinnedBatchRing ships the order and the offsets in that order with the batch
                offs64, order = batch.get("offsets"), batch.get("order")
                if offs64 is None:
                    order, offs64 = ops.length_order(lens)
                    order, offs64 = order.to(dev), offs64.to(dev)
                packed = ops.pack_rows(hist.to(dev, torch.int64).contiguous(), pos, neg, offs64, rows, order=order)
                packed |= {"batch": hist.shape[0], "seq_len": L}
                pos, neg = packed["pos"], packed["neg"]
                padded_positions = hist.shape[0] * L
        opts = dict(train_head=c.train_loss, all_heads=c.log_all_losses, mask_false_negatives=c.mask_false_negatives,
                    mode=N.NEG_CATALOG if catalogue else N.NEG_SHARED, scale=c.scale, margin=c.margin, precision=c.precision,
                    table_bf16=m.table_bf16, num_hard_negatives=c.num_hard_negatives, padded_positions=padded_positions)
        overlap = (defer_logging and c.log_all_losses and m.flat.requires_grad and torch.is_grad_enabled()
                   and m.table_bf16 is not None and c.precision == "bf16" and c.num_hard_negatives == 0
                   # unmasked InfoNCE: ONE call runs the logging pass first and pins the gradient pass's running
                   # maximum from its records (lean epilogue) -- worth more than the overlap
                   and not (c.train_loss == "InfoNCELoss" and not c.mask_false_negatives))
        prep = None
        if overlap:
            # The index-only half of both loss calls (query compaction, multiplicities, distinct negatives: 7 small launches
            # each) needs the key mask, not the token embeddings: it runs on the side stream underneath the forward, behind an
            # event the encoder forward records right after the launch that writes the mask -- instead of 40 us of small
            # launches between the forward's last kernel and the loss kernels.
            H, n_rows, T = m.config.hidden_size, m.embeddings.shape[0], (pos.numel() if packed else hist.shape[0] * L)
            if getattr(self, "_ev_embed", None) is None:
                self._ev_embed, self._ev_prep = torch.cuda.Event(), torch.cuda.Event()
                self._ev_embed.record()  # (creates the handle)
            ws_log = ops.sampled_loss_workspace(m.flat, T, H, n_rows, **opts)
            ws_grad = ops.sampled_loss_workspace(m.flat, T, H, n_rows, **opts)
            prep = (ws_log, ws_grad, H, n_rows)
        # (the forward records the event right after the launch that writes the key mask: xfmr_encoder_cfg.embed_event)
        tok, key_mask = m._encode_tokens(None if packed else hist, embed_event=self._ev_embed.cuda_event if overlap else None,
                                         packed=packed)
        if m.config.is_normalized:  # models.py:393-394: the queries are the L2-normalised token embeddings
            tok = ops.l2_normalize(tok)
        assert packed is not None or tok.shape[1] == L
        if overlap:
            # The six logging heads + statistics do not feed the gradient: evaluate them on a side stream so the
            # (VALU-bound) logging pass runs underneath the (latency-bound) encoder backward. The caller joins
            # with sync_logging() before reading them (Trainer.fit_step / bench.py do, after optimizer.step()).
            main = torch.cuda.current_stream()
            side = self._side_stream()
            ws_log, ws_grad, H, n_rows = prep
            side.wait_event(self._ev_embed)  # the key mask exists (the forward is still running)
            with torch.cuda.stream(side):
                for ws, heads in ((ws_log, 2), (ws_grad, False)):
                    ops.sampled_loss_prepare(ws, key_mask, pos, neg, m.table_rnorm, n_rows, H, **(opts | {"all_heads": heads}))
                d_tok0 = torch.zeros_like(tok)  # the gradient buffer, zeroed here instead of in front of the gradient pass
                self._ev_prep.record(side)
            # enqueued BEFORE the gradient pass: it needs the forward's output only, and at its lowest priority it takes
            # what the gradient pass (800 workgroups on 512 slots: 1.56 rounds) leaves idle, then the encoder backward's gaps
            side.wait_stream(main)
            with torch.cuda.stream(side):
                # all_heads=2: every head except the train head (its value comes from the launch below)
                losses, stats, _ = ops.sampled_loss(
                    tok.detach(), key_mask, pos, neg, m.embeddings, m.table_rnorm, need_grad=False,
                    **(opts | {"all_heads": 2, "workspace": ws_log, "prepared": True, "profile_log": profile_log})
                )
            main.wait_event(self._ev_prep)  # (recorded long ago)
            train_loss, losses_train, stats_t = ops.sampled_loss_train(
                tok, key_mask, pos, neg, m.embeddings, m.table_rnorm,
                opts | {"all_heads": False, "workspace": ws_grad, "prepared": True, "d_tok_zeroed": d_tok0,
                        "profile_grad": profile_grad}
            )
            for tns in (ws_log, ws_grad):
                tns.record_stream(side)
            d_tok0.record_stream(main)  # allocated on the side stream, used on the main one
            for tns in (tok, key_mask, pos, neg):
                if tns is not None:
                    tns.record_stream(side)
            self._logging_pending = True
        else:
            train_loss, losses, stats = ops.sampled_loss_train(
                tok, key_mask, pos, neg, m.embeddings, m.table_rnorm,
                opts | {"profile_grad": profile_grad, "profile_log": profile_log}
            )
        out: dict = {}
        n_query = stats[N.STAT["n_query"]]
        names = [cls.__name__ for cls in LOSS_CLASSES]
        for i, name in enumerate(names):
            if not c.log_all_losses and name != c.train_loss:
                continue
            val = train_loss if name == c.train_loss else losses[i]
            out[f"loss/{name}"] = val
            if not overlap:
                out[f"loss/{name}Mean"] = losses[N.NUM_LOSSES + i]  # computed by the final kernel
        if overlap:  # raw device vectors: the side-stream pass (train head entries = 0) and the gradient call (train head only)
            out["losses/device"] = losses
            out["losses_train/device"] = losses_train
        batch_size, seq_len = hist.shape[0], L  # (the padded shape, also when the rows were packed)
        numel = batch_size * seq_len
        # trainer.py:241-244: known without a device sync, logged on every path
        out |= {"batch/size": batch_size, "batch/seq_len": seq_len, "batch/numel": numel}
        if sync_metrics and not overlap:
            s = stats.tolist()  # one device->host sync (the reference does 11 .item() calls)
            attn_nz, pos_nz = int(s[N.STAT["n_valid"]]), int(s[N.STAT["n_query"]])
            out |= {
                "batch/attention_non_zero": attn_nz, "batch/attention_density": attn_nz / (numel + 1e-9),
                "batch/positive_non_zero": pos_nz, "batch/positive_density": pos_nz / (attn_nz + 1e-9),
            }
            if c.log_all_losses:
                out |= stats_to_dict(s)
        else:
            out["stats/device"] = stats
        return out

    def logged_values(self, out: dict) -> dict[str, float]:
        """Host-side view of a ``compute_losses(..., defer_logging=True)`` result: joins the side stream and
        returns every ``loss/<Class>``, ``loss/<Class>Mean``, ``batch/*`` and ``logits/*`` entry as floats
        (one device->host sync), exactly the keys the reference logs (``trainer.py:241-263``)."""
        self.sync_logging()
        c = self.config
        if "stats/device" not in out:
            return {k: float(v) for k, v in out.items() if not k.endswith("/device")}
        s = out["stats/device"].tolist()
        attn_nz, pos_nz = int(s[N.STAT["n_valid"]]), int(s[N.STAT["n_query"]])
        res: dict[str, float] = {}
        dev = out.get("losses/device")
        vals = dev.tolist() if dev is not None else None
        for i, cls in enumerate(LOSS_CLASSES):
            k = f"loss/{cls.__name__}"
            if k not in out:
                continue
            v = float(out[k]) if (vals is None or cls.__name__ == c.train_loss) else vals[i]
            res[k] = v
            res[k + "Mean"] = v / (pos_nz + 1e-9)
        res |= {k: out[k] for k in ("batch/size", "batch/seq_len", "batch/numel") if k in out}
        res |= {
            "batch/attention_non_zero": attn_nz, "batch/positive_non_zero": pos_nz,
            "batch/positive_density": pos_nz / (attn_nz + 1e-9),
        }
        if "batch/numel" in out:
            res["batch/attention_density"] = attn_nz / (out["batch/numel"] + 1e-9)
        if c.log_all_losses:
            res |= stats_to_dict(s)
        return res

    # ------------------------------------------------------------------ validation path (trainer.py:186-325)
    @property
    def items_index(self):
        """Exact item index over the model's frozen table (the reference builds a LanceDB ANN index at
        ``on_validation_start``, ``trainer.py:316-325``; the table here never changes, so the index is a view)."""
        from .retrieval import ExactItemIndex

        assert self.model is not None and self.model.embeddings is not None
        idx = getattr(self, "_items_index", None)
        if idx is None or idx.table.data_ptr() != self.model.embeddings.data_ptr():
            idx = ExactItemIndex(self.model.embeddings, self.model.table_rnorm)
            self._items_index = idx
        return idx

    def _to_idx(self, item_ids) -> list[int]:
        """Item ids -> table rows; unknown ids are dropped (``models.py:347-364``). Integer inputs are rows already."""
        m = self.model
        if len(item_ids) and not isinstance(item_ids[0], (str, bytes)):
            return [int(i) for i in item_ids]
        assert m.id2idx is not None, "configure_embeddings(items_dataset) provides the id -> row mapping"
        if hasattr(m.id2idx, "index"):
            return [int(m.id2idx[i]) for i in item_ids if i in m.id2idx.index]
        return [int(m.id2idx[i]) for i in item_ids if i in m.id2idx]

    @torch.no_grad()
    def recommend(self, item_ids, *, top_k: int = 0, exclude_item_ids=None):
        """``trainer.py:186-211``: nearest items to the pooled embedding of ``item_ids``, ``exclude_item_ids`` left
        out. Returns ``{"item_idx": (k,) int64 rows of the table (-1 = fewer than k left), "score": (k,)}``."""
        hist = self._to_idx(item_ids)
        emb = self.model(torch.as_tensor(hist, dtype=torch.int64, device=self.model.device)[None, :])["sentence_embedding"]
        excl = None if exclude_item_ids is None else [self._to_idx(exclude_item_ids)]
        idx, score = self.items_index.search(emb, excl, top_k=top_k or self.config.top_k)
        return {"item_idx": idx[0], "score": score[0]}

    def predict_step(self, row):
        """``trainer.py:305-314``: recommendations for the row's history, the history itself excluded."""
        hist = list(row["history"]["item_id"])
        return self.recommend(hist, top_k=self.config.top_k, exclude_item_ids=hist)

    def compute_metrics(self, row, stage: str = "val") -> dict[str, torch.Tensor]:
        """``trainer.py:266-286``: the seven retrieval metrics of one validation row under ``{stage}/<name>``."""
        from .retrieval import compute_retrieval_metrics

        recs = self.predict_step(row)
        tgt_ids = [i for i, l in zip(row["target"]["item_id"], row["target"]["label"]) if l]
        metrics = compute_retrieval_metrics(recs["item_idx"], self._to_idx(tgt_ids), top_k=self.config.top_k)
        return {f"{stage}/{k}": v for k, v in metrics.items()}

    def validation_step(self, row, batch_idx: int = 0):
        metrics = self.compute_metrics(row, stage="val")
        self.log_dict(metrics, batch_size=1)
        return metrics

    def test_step(self, row, batch_idx: int = 0):
        metrics = self.compute_metrics(row, stage="test")
        self.log_dict(metrics, batch_size=1)
        return metrics

    def training_step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        """``trainer.py:288-291``. Runs the step the benchmark measures: metrics stay on the device (no host sync), the
        six logging heads + statistics go to the lowest-priority side stream underneath the backward. The train loss is
        logged here; everything else is logged by :meth:`on_train_batch_end` -- after the side stream has been joined --
        as 0-dim DEVICE tensors (Lightning's ``log_dict`` takes them and converts at its own logging interval)."""
        prof = getattr(self, "profile_events", None) or (None, None)  # bench.py: hipEvent pairs around the two loss passes
        defer = getattr(self, "defer_logging", "auto")
        if defer == "auto":
            # the side-stream logging pass pays once it is long enough to be worth hiding: B x L >= 51 200 tokens
            # (round 4, bench.py --overlap on / off, one box: batch 256 x 200 1.918-1.947 against 1.958-1.962 ms, batch 128
            #  1.193 against 1.178; round 1 had measured -1.4 % at 256 and set 102 400)
            h = batch["history_item_idx"]
            defer = h.shape[0] * min(h.shape[1], self.model.max_seq_length) >= 51200
        out = self.compute_losses(batch, sync_metrics=False, defer_logging=bool(defer),
                                  profile_grad=prof[0], profile_log=prof[1])
        key = f"loss/{self.config.train_loss}"
        # only DETACHED tensors outlive the step: a retained train loss would keep the step's autograd graph -- the saved
        # activations and `flat`'s AccumulateGrad node with the stream it was created on -- alive into the next step (and
        # into a later hipGraph capture on another stream: ADVICE r3)
        self._pending_out = {k: (v.detach() if (torch.is_tensor(v) and v.requires_grad) else v) for k, v in out.items()}
        self.logged = {}
        self.log_dict({key: out[key].detach()})
        return out[key]

    def backward(self, loss: torch.Tensor, *args, **kwargs) -> None:
        """Lightning's ``LightningModule.backward`` hook (called by its automatic optimisation; ``Trainer.fit_step`` and
        ``bench.py`` call it too): ``loss.backward()`` with the persistent unit gradient -- no ones-fill launch, and the fused
        loss skips its multiply by 1 (ops.unit_grad)."""
        if loss.dim() == 0 and loss.is_cuda and not args and not kwargs:
            loss.backward(gradient=ops.unit_grad(loss))
        else:
            loss.backward(*args, **kwargs)

    def on_train_batch_end(self, outputs=None, batch=None, batch_idx: int = 0) -> None:
        """Lightning hook (after ``optimizer.step``): join the logging stream and log the step's remaining values."""
        out = getattr(self, "_pending_out", None)
        if out is None:
            return
        self._pending_out = None
        self.last_out = out  # the step's raw device outputs (bench.py reads the statistics vector from it)
        self.sync_logging()
        self.log_dict(self.device_log_dict(out))

    def device_log_dict(self, out: dict) -> dict[str, torch.Tensor]:
        """Every key the reference logs (``trainer.py:241-263``, ``losses.py:392-404``) as 0-dim device tensors -- views
        of the kernels' output vectors, no host sync, no torch arithmetic. Call after :meth:`sync_logging`.
        (``logits/{pos,neg}/*`` of an empty selection are NaN here where the reference omits the key.)"""
        c = self.config
        res: dict[str, torch.Tensor] = {}
        stats = out.get("stats/device")
        if stats is None:  # a synchronous compute_losses result: already complete
            return {k: torch.as_tensor(v) for k, v in out.items() if not k.endswith("/device")}
        side, own = out.get("losses/device"), out.get("losses_train/device")
        for i, cls in enumerate(LOSS_CLASSES):
            k = f"loss/{cls.__name__}"
            if k not in out:
                continue
            is_train = cls.__name__ == c.train_loss
            res[k] = out[k].detach()
            if k + "Mean" in out:
                res[k + "Mean"] = out[k + "Mean"]
            elif side is not None:
                res[k + "Mean"] = (own if is_train and own is not None else side)[N.NUM_LOSSES + i]
        for k in ("batch/size", "batch/seq_len", "batch/numel"):  # host constants (trainer.py:241-244)
            if k in out:
                res[k] = torch.as_tensor(out[k])
        res["batch/attention_non_zero"] = stats[N.STAT["n_valid"]]
        res["batch/positive_non_zero"] = stats[N.STAT["n_query"]]
        res["batch/attention_density"] = stats[N.STAT["attn_density"]]
        res["batch/positive_density"] = stats[N.STAT["pos_density"]]
        if c.log_all_losses:
            res["logits/neg/density"] = stats[N.STAT["neg_density"]]
            for side_ in ("pos", "neg"):
                for kk in ("mean", "std", "min", "max"):
                    res[f"logits/{side_}/{kk}"] = stats[N.STAT[f"{side_}_{kk}"]]
        return res

    def log_dict(self, d, *a, **k):  # noqa: D401 - lightning API
        if _lp is not None and getattr(self, "_trainer", None) is not None:  # pragma: no cover
            return super().log_dict(d, *a, **k)
        # plain-module build: the step's values accumulate in `logged` (training_step starts a fresh dict)
        self.logged.update(d)
        return None

    def configure_optimizers(self) -> torch.optim.Optimizer:
        """``trainer.py:327-332`` with the fused kernel."""
        return FusedAdamW(self.parameters(), lr=self.config.learning_rate, weight_decay=self.config.weight_decay)

    @property
    def example_input_array(self):
        return (torch.as_tensor([[0], [1]], device=self._device()),)

    def state_dict(self, *args, **kwargs):
        """HF-keyed encoder tensors under the reference's prefix; the frozen table is omitted
        (``trainer.py:352-362``)."""
        sd = super().state_dict(*args, **kwargs)
        sd.pop("model.embeddings.weight", None)
        if self.model is not None:
            sd.pop("model.flat", None)
            for k, v in self.model.encoder_state_dict().items():
                sd[f"model.model.0.auto_model.{k}"] = v
        return sd

    _ENC_PREFIX = "model.model.0.auto_model."
    # tensors a reference checkpoint carries that this build has no use for (never trained on this path: SURVEY F12)
    _IGNORED_ENC_KEYS = ("embeddings.word_embeddings.weight", "embeddings.position_ids", "embeddings.token_type_ids",
                         "pooler.dense.weight", "pooler.dense.bias")

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        """Inverse of :meth:`state_dict`: HF-keyed encoder tensors under ``model.model.0.auto_model.`` -- this build's
        checkpoints and the reference's (``trainer.py:352-362`` only pops the table) -- go back into the flat buffer.
        Returns torch's ``(missing_keys, unexpected_keys)``; ``strict=True`` raises on either, as ``nn.Module`` does.
        Lightning resumes with ``strict_loading=False`` (``trainer.py:129``), i.e. ``strict=False``."""
        from torch.nn.modules.module import _IncompatibleKeys

        if self.model is None:
            self.configure_model()
        pre = self._ENC_PREFIX
        enc = {k[len(pre):]: v for k, v in state_dict.items() if k.startswith(pre)}
        views = self.model.encoder_state_dict()
        if "model.flat" in state_dict and not enc:  # a raw dump of the flat buffer
            flat = torch.as_tensor(state_dict["model.flat"])
            if flat.shape != self.model.flat.shape:
                raise RuntimeError(f"model.flat: shape {tuple(flat.shape)} != {tuple(self.model.flat.shape)}")
            with torch.no_grad():
                self.model.flat.copy_(flat.to(self.model.flat.device, torch.float32))
            enc = {k: None for k in views}
        missing = [pre + k for k in views if k not in enc]
        unexpected = [pre + k for k in enc if k not in views and k not in self._IGNORED_ENC_KEYS]
        unexpected += [k for k in state_dict if not k.startswith(pre) and k not in ("model.flat", "model.embeddings.weight")]
        bad_shape = [pre + k for k, v in enc.items() if k in views and v is not None
                     and tuple(torch.as_tensor(v).shape) != tuple(views[k].shape)]
        if bad_shape:
            raise RuntimeError(f"size mismatch for {bad_shape[:4]}")
        if strict and (missing or unexpected):
            raise RuntimeError(f"Error(s) in loading state_dict: missing keys {missing[:4]}{'...' if len(missing) > 4 else ''}, "
                               f"unexpected keys {unexpected[:4]}{'...' if len(unexpected) > 4 else ''}")
        self.model.load_encoder_state_dict({k: v for k, v in enc.items() if v is not None and k in views}, strict=False)
        if "model.embeddings.weight" in state_dict:  # a checkpoint that kept the table (not the reference's): install it
            self.model.set_table(torch.as_tensor(state_dict["model.embeddings.weight"]).to(self.model.device, torch.float32))
        return _IncompatibleKeys(missing, unexpected)

    def save(self, path) -> None:
        assert self.model is not None
        self.model.save(str(path))


class Trainer:
    """Minimal stand-in for Lightning's automatic optimisation (``zero_grad -> training_step -> backward ->
    [all-reduce] -> optimizer.step``), single process or one process per GPU (``torch.distributed``)."""

    def __init__(self, module: RecommenderLightningModule, *, world_size: int = 1, process_group=None):
        self.module = module
        module.configure_model()
        self.optimizer = module.configure_optimizers()
        self.world_size = world_size
        self.process_group = process_group
        self.exchange = None
        self.default_stream_steps = 0  # eager steps this trainer ran on the device's DEFAULT stream
        if world_size > 1:
            import os

            for g in self.optimizer.param_groups:
                g["grad_scale"] = 1.0 / world_size  # DDP averages gradients: SUM all-reduce then / W
            # One message behind the backward by default. XFMR_ALLREDUCE_HALVES=1: two halves, the upper layers' underneath
            # the lower layers' backward (distributed.HalvedAllReduce) -- its event ordering is tested on one GPU
            # (tests/test_gpu_ddp.py), but RCCL's transport has never run under it (no multi-GPU box was available to the
            # build: DESIGN.md section 6), so it stays opt-in until a scaling record exists.
            if os.environ.get("XFMR_ALLREDUCE_HALVES", "0") == "1" and os.environ.get("XFMR_ALLREDUCE_SINGLE", "0") != "1":
                from .distributed import HalvedAllReduce

                self.exchange = HalvedAllReduce(module.model, process_group)
            elif os.environ.get("XFMR_ALLREDUCE", "") == "abi" and module.model.flat.is_cuda:
                # the exchange through the C ABI (xfmr_allreduce_flat: RCCL underneath, no torch collective in the step)
                from .distributed import AbiAllReduce

                self.exchange = AbiAllReduce(module.model.device, process_group)

    def allreduce_(self, flat_grad: torch.Tensor) -> None:
        """The step's one exchange: SUM of the flat gradient over the ranks (1 / W is folded into AdamW); with
        XFMR_ALLREDUCE_HALVES=1 in two halves, the upper layers' underneath the rest of the backward."""
        if self.world_size <= 1:
            return
        if self.exchange is not None:
            self.exchange.reduce_(flat_grad)
        else:
            from .distributed import allreduce_flat_grad_

            allreduce_flat_grad_(flat_grad, self.process_group)

    def fit_step(self, batch) -> torch.Tensor:
        """One step through the module's Lightning seam, in Lightning's order (``zero_grad -> training_step -> backward
        -> [all-reduce] -> optimizer.step -> on_train_batch_end``): what ``bench.py`` times."""
        m = self.module
        m.train()
        dev = m.model.device
        if dev.type == "cuda" and not torch.cuda.is_current_stream_capturing() \
                and torch.cuda.current_stream(dev) == torch.cuda.default_stream(dev):
            self.default_stream_steps += 1  # (a later hipGraph capture refuses: GraphedStep)
        self.optimizer.zero_grad(set_to_none=True)
        loss = m.training_step(batch, 0)
        m.backward(loss)
        self.allreduce_(m.model.flat.grad)
        self.optimizer.step()
        m.on_train_batch_end(loss, batch, 0)  # joins the logging stream (leaving the pass to finish underneath the next
        return loss.detach()                  # step's forward measured no gain: 3.411 vs 3.414 ms)

    def fit(self, batches, max_steps: int | None = None, *, ring_slots: int = 6, graph: str = "off",
            graph_probe_steps: int = 20) -> list[float]:
        """Steps over an iterable of collated batches. Batches that arrive in HOST memory (the reference's DataLoader
        output, ``data.py:915-927``) are handed over through a :class:`~xfmr_rec_amd.data.PinnedBatchRing`: the copy of
        batch i + 1 runs underneath step i; device-resident batches (``DeviceSeqDataset.sample_batch``) are used as they are.

        ``graph``: "off" -- every step is enqueued launch by launch; "on" -- after three eager steps (they create every
        lazily made object) the step is captured once as a hipGraph (:class:`GraphedStep`) and replayed for every batch
        of the captured shape; "auto" -- ``graph_probe_steps`` eager steps are timed, then as many replays, and the faster
        form runs the rest (replay wins where the host's launch rate bounds the step -- BASELINE config 1's shape: 0.31
        against 0.48 ms -- and changes nothing where the GPU does: DESIGN.md section 5). Every probe step is a real
        training step on its own batch. Batches of another shape (a short last batch) take the eager step. Single
        process only; ``self.graph_choice`` records what ran."""
        from .data import SEQ_BATCH_KEYS, PinnedBatchRing

        if graph not in ("off", "on", "auto"):
            raise ValueError(f"graph must be 'off', 'on' or 'auto', got {graph!r}")
        if self.world_size != 1:
            graph = "off"  # (the all-reduce of a data-parallel step is not captured)
        defer_before = getattr(self.module, "defer_logging", "auto")
        if graph != "off":
            _refuse_capture_after_default_stream_steps(self)
            # the step counter the captured kernels read (dropout stream, AdamW bias corrections) lives in HBM from the
            # first step on, so that the eager steps in front of the capture and the replays after it are one sequence
            mdl = self.module.model
            if getattr(mdl, "step_device", None) is None:
                mdl.use_device_step(True)
                done = max((st.get("step", 0) for st in self.optimizer.state.values()), default=0)
                mdl.step_device.fill_(int(done))
            self.optimizer.step_device = mdl.step_device
            if getattr(self.module, "defer_logging", "auto") == "auto":
                self.module.defer_logging = False  # one stream: what the capture records (GraphedStep)
        gstep, use_graph = None, False
        probe: dict = {}
        self.graph_choice = "eager"
        n_warm = 3

        def _sync_time():
            torch.cuda.synchronize(self.module.model.device)
            return time.perf_counter()

        out = []
        t0 = time.time()
        ring = None
        # torch's capture protocol: the eager steps in front of a capture run on a NON-default stream (autograd's
        # AccumulateGrad nodes are bound to the stream of their first backward; captured from the default stream the
        # capture faults). The whole loop therefore runs on a side stream when a graph may be captured.
        dev = self.module.model.device
        cur = torch.cuda.current_stream(dev) if graph != "off" else None
        side = torch.cuda.Stream(device=dev) if graph != "off" else None
        if side is not None:
            side.wait_stream(cur)
        with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
            it = iter(batches)
            nxt = next(it, None)
            i = 0
            while nxt is not None and (max_steps is None or i < max_steps):
                b = nxt
                on_host = not b[SEQ_BATCH_KEYS[0]].is_cuda
                if on_host:
                    if ring is None or b[SEQ_BATCH_KEYS[0]].shape[0] > ring.shape[1] or b[SEQ_BATCH_KEYS[0]].shape[1] > ring.shape[2]:
                        if ring is not None:
                            ring.close()
                        bs, width = b[SEQ_BATCH_KEYS[0]].shape
                        ring = PinnedBatchRing(self.module.model.device, bs, max(width, self.module.model.max_seq_length),
                                               slots=ring_slots)
                    if ring.pending == 0:
                        ring.stage(b)
                    dev_b = ring.take()
                    if graph != "off":
                        # a captured step has fixed launch sizes: it runs the padded layout, and so do the eager steps around
                        # it (one sequence of steps, one layout) -- the rows' lengths are not passed on
                        dev_b = {k: dev_b[k] for k in SEQ_BATCH_KEYS}
                else:
                    dev_b = b
                nxt = next(it, None)
                if nxt is not None and ring is not None and not nxt[SEQ_BATCH_KEYS[0]].is_cuda \
                        and nxt[SEQ_BATCH_KEYS[0]].shape[0] <= ring.shape[1] and nxt[SEQ_BATCH_KEYS[0]].shape[1] <= ring.shape[2]:
                    ring.stage(nxt)  # in flight while this step computes
                if graph != "off" and i >= n_warm:
                    if graph == "auto" and "eager_t0" not in probe:
                        probe["eager_t0"], probe["eager_i0"] = _sync_time(), i
                    eager_done = graph == "on" or i - probe["eager_i0"] >= graph_probe_steps
                    if gstep is None and eager_done:
                        if graph == "auto":
                            probe["eager_ms"] = (_sync_time() - probe["eager_t0"]) / (i - probe["eager_i0"]) * 1e3
                        gstep = GraphedStep(self, dev_b, warmup=0)
                        use_graph = True
                        if graph == "auto":
                            probe["graph_t0"], probe["graph_i0"] = _sync_time(), i
                    elif graph == "auto" and gstep is not None and "graph_ms" not in probe \
                            and i - probe["graph_i0"] >= graph_probe_steps:
                        probe["graph_ms"] = (_sync_time() - probe["graph_t0"]) / (i - probe["graph_i0"]) * 1e3
                        use_graph = probe["graph_ms"] < probe["eager_ms"]
                    self.graph_choice = "graph" if use_graph else "eager"
                if use_graph and gstep.matches(dev_b):
                    out.append(gstep(dev_b).clone())  # (the captured loss tensor is overwritten by the next replay)
                else:
                    out.append(self.fit_step(dev_b))
                i += 1
        if side is not None:
            cur.wait_stream(side)
        if graph != "off":
            self.module.defer_logging = defer_before  # (the capture needed one stream; the caller's setting comes back)
            self.optimizer.sync_step_from_device()  # replays advanced the counter on the device only
        self.graph_probe = {k: round(v, 4) for k, v in probe.items() if k.endswith("_ms")}
        if ring is not None:
            ring.release()
            ring.close()
        self.elapsed = time.time() - t0
        return [float(v) for v in out]  # (one host sync at the end, not one per step)


def _refuse_capture_after_default_stream_steps(trainer: "Trainer") -> None:
    """torch's capture protocol: autograd's AccumulateGrad node of a parameter is bound to the stream of the first backward
    that used it and synchronises with that stream from then on; eager steps on the device's DEFAULT (legacy) stream
    followed by a capture on a side stream made the capture fault (scripts/probe/fit_graph_probe.py, round 3). A clear
    Python error instead of that fault."""
    if getattr(trainer, "default_stream_steps", 0) > 0:
        raise RuntimeError(
            f"this Trainer has run {trainer.default_stream_steps} eager step(s) on the device's default stream; a hipGraph "
            "capture after that is refused (torch's capture protocol: warm-up steps must run on a side stream -- "
            "`with torch.cuda.stream(torch.cuda.Stream()): trainer.fit_step(...)`, or let Trainer.fit(graph=...) / "
            "GraphedStep(warmup=...) run them). Build a new Trainer for the captured run.")


class GraphedStep:
    """One training step of ``trainer`` (``Trainer.fit_step``: zero_grad -> training_step -> backward -> AdamW ->
    on_train_batch_end) captured ONCE as a hipGraph for a fixed batch shape and replayed per step: the ~80 kernel
    launches of a step become one graph launch. For small batches the eager step is bound by the host's launch rate,
    not by the GPU (batch 32 at config 2: 0.84 ms per step against 0.8-0.95 ms of enqueueing).

    What changes from step to step is read from DEVICE memory inside the captured kernels -- the dropout stream
    (``xfmr_encoder_cfg.step_device``) and AdamW's step count (``xfmr_adamw_dev``), both the model's ``step_device``
    counter, advanced by the step's last launch (``xfmr_step_advance``) -- so every replay draws new masks and applies the
    right bias corrections: replay == the eager step with the same counter, bit for bit (tests/test_gpu_graph.py).

        step = GraphedStep(trainer, example_batch)     # device tensors of the shape every batch will have
        loss = step(batch)                             # copies the indices into the captured buffers, replays

    Single process only (the all-reduce of a data-parallel step is not captured here). The eager steps in front of the
    capture must NOT have run on the default stream (torch's capture protocol; the capture faults otherwise:
    scripts/probe/fit_graph_probe.py) -- the warm-up here and ``Trainer.fit(graph=...)`` both use a side stream."""

    def __init__(self, trainer: Trainer, example_batch: dict, warmup: int = 3, overlap: bool = False):
        """``overlap=True`` captures the step with its two forks at ANY size -- the weight-gradient GEMMs on the context's
        side stream (``XFMR_ENC_DW_SIDE_ANY``) and the logging heads on the logging stream (``defer_logging``) -- as
        branches of the graph. Same bits (tests/test_gpu_graph.py), but on this runtime (ROCm 7.2 / torch 2.10) a graph
        with cross-stream branches replays 2-3x SLOWER than the single-stream capture (config 2 at batch 32: 2.07 ms
        against 0.85; H 64 / L 50 / batch 64: 1.39 against 0.31), so the default is the single-stream step."""
        from .data import SEQ_BATCH_KEYS

        m = trainer.module
        if trainer.world_size != 1:
            raise ValueError("GraphedStep captures a single-process step")
        _refuse_capture_after_default_stream_steps(trainer)
        self.trainer, self.keys = trainer, SEQ_BATCH_KEYS
        dev = m.model.device
        if overlap:
            m.model.enc_flags = getattr(m.model, "enc_flags", 0) | N.ENC_DW_SIDE_ANY
            m.defer_logging = True
        elif getattr(m, "defer_logging", "auto") == "auto":
            m.defer_logging = False  # one stream inside the capture
        if getattr(m.model, "step_device", None) is None:
            # the counter continues the optimizer's completed steps (AdamW's bias corrections and the dropout stream go on
            # from where the eager steps stopped, as Trainer.fit seeds it)
            m.model.use_device_step(True)
            done = max((st.get("step", 0) for st in trainer.optimizer.state.values()), default=0)
            m.model.step_device.fill_(int(done))
        trainer.optimizer.step_device = m.model.step_device
        m.train()
        m._pending_out = None  # nothing of an earlier step (its tensors, their autograd graph) reaches into the capture
        m.last_out = None
        self.static = {k: example_batch[k].to(dev, torch.int64).clone() for k in self.keys}
        # eager warm-up on a side stream (torch's capture protocol): creates every lazily made object -- optimizer state,
        # the model's xfmr_context, allocator pools -- so that the capture itself creates nothing. ``warmup=0``: the caller
        # has already run eager steps of this shape (Trainer.fit does, on real batches).
        if warmup > 0:
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(warmup):
                    trainer.fit_step(self.static)
            torch.cuda.current_stream().wait_stream(side)
        trainer.optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss = trainer.fit_step(self.static)
        self.logged = dict(m.logged)  # 0-dim device tensors inside the graph's pool: refreshed by every replay

    def matches(self, batch: dict) -> bool:
        """Whether ``batch`` has the captured shape (a replay needs it)."""
        return all(tuple(batch[k].shape) == tuple(self.static[k].shape) for k in self.keys)

    def __call__(self, batch: dict) -> torch.Tensor:
        for k in self.keys:
            self.static[k].copy_(batch[k], non_blocking=True)
        self.graph.replay()
        return self.loss
