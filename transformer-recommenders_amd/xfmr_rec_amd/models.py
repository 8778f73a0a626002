"""``RecommenderModel``: host-side mirror of the reference's ``xfmr_rec/models.py`` over the HIP encoder.

Keeps the reference's surface -- ``ModelConfig`` fields/defaults (``models.py:22-48``),
``configure_embeddings``, ``forward(item_idx=None, *, item_embeds=None)``, ``compute_embeds``, ``encode``,
``save`` / ``load``, properties ``device`` / ``max_seq_length`` (``models.py:176-419``) -- while the compute
(gather + mask + BERT stack, fwd and bwd) runs in ``libxfmr_hip.so``.

Differences that are deliberate and visible:

* The encoder is not a HuggingFace/sentence-transformers object. All trainable tensors live in one flat
  fp32 ``nn.Parameter`` (``self.flat``); ``encoder_state_dict()`` exposes them under the HF BERT key names
  the reference checkpoints use, and ``load_encoder_state_dict()`` takes the same.
  ``word_embeddings`` / ``pooler`` never receive a gradient in the reference and are not created.
* Nothing is fetched from the network. ``None`` fields of ``ModelConfig`` (the reference's own ``config.yaml``
  leaves ``hidden_size: null``) are resolved offline: from the published config of ``pretrained_model_name`` when it
  is a known model (``params.KNOWN_PRETRAINED`` -- what the reference's ``AutoModel.from_pretrained`` would return,
  ``models.py:69-91``), else ``hidden_size = 32 x num_attention_heads`` (the head size of the kernels); the item
  table's width is checked against it in ``configure_embeddings``.
* ``compute_embeds`` returns the candidates structured (:class:`~xfmr_rec_amd.losses.SharedNegatives`)
  rather than as the materialised ``(Np, 1+N, H)`` tensor.
"""

from __future__ import annotations

import os

import json
import pathlib
from typing import Literal

import pydantic
import torch

from . import _native as N
from . import ops
from .losses import SharedNegatives
from .params import (
    ATTENTION_PROBS_DROPOUT_PROB,
    HIDDEN_DROPOUT_PROB,
    INITIALIZER_RANGE,
    KNOWN_PRETRAINED,
    LAYER_NORM_EPS,
    PRETRAINED_MODEL_NAME,
)


class ModelConfig(pydantic.BaseModel):
    """Same fields and defaults as the reference (``models.py:22-48``)."""

    vocab_size: int | None = 1
    hidden_size: int | None = None
    num_hidden_layers: int | None = 1
    num_attention_heads: int | None = 12
    intermediate_size: int | None = 48
    max_seq_length: int | None = 32
    is_decoder: bool = True

    pretrained_model_name: str = PRETRAINED_MODEL_NAME
    pooling_mode: Literal["mean", "max", "cls", "lasttoken"] = "mean"
    is_normalized: bool = False


def encoder_param_names(num_layers: int) -> list[str]:
    """HF BERT state_dict keys in the order of the flat buffer (``include/xfmr_hip.h``)."""
    names = [
        "embeddings.position_embeddings.weight",
        "embeddings.token_type_embeddings.weight",
        "embeddings.LayerNorm.weight",
        "embeddings.LayerNorm.bias",
    ]
    for i in range(num_layers):
        p = f"encoder.layer.{i}."
        names += [p + f"attention.self.{n}.weight" for n in ("query", "key", "value")]
        names += [p + f"attention.self.{n}.bias" for n in ("query", "key", "value")]
        names += [
            p + "attention.output.dense.weight", p + "attention.output.dense.bias",
            p + "attention.output.LayerNorm.weight", p + "attention.output.LayerNorm.bias",
            p + "intermediate.dense.weight", p + "intermediate.dense.bias",
            p + "output.dense.weight", p + "output.dense.bias",
            p + "output.LayerNorm.weight", p + "output.LayerNorm.bias",
        ]
    return names


def encoder_param_shapes(H: int, I: int, max_pos: int, num_layers: int) -> list[tuple[int, ...]]:
    shapes: list[tuple[int, ...]] = [(max_pos, H), (2, H), (H,), (H,)]
    for _ in range(num_layers):
        shapes += [(H, H)] * 3 + [(H,)] * 3 + [(H, H), (H,), (H,), (H,), (I, H), (I,), (H, I), (H,), (H,), (H,)]
    return shapes


def flat_layout(H: int, I: int, max_pos: int, num_layers: int):
    """(names, shapes, offsets, total): pure-Python twin of ``xfmr_param_offsets`` (checked in the tests)."""
    names = encoder_param_names(num_layers)
    shapes = encoder_param_shapes(H, I, max_pos, num_layers)
    offsets, o = [], 0
    for s in shapes:
        offsets.append(o)
        n = 1
        for d in s:
            n *= d
        o += n
    return names, shapes, offsets, o


class RecommenderModel(torch.nn.Module):
    def __init__(
        self,
        config: ModelConfig,
        *,
        device: torch.device | str | None = None,
        model=None,
        precision: str = "bf16",
        seed: int = 0,
    ) -> None:
        """``model`` (a pre-built sentence-transformers object in the reference, ``models.py:177-199``) may be
        an HF-keyed state dict here; ``precision`` selects the MFMA arithmetic ("bf16" or "fp32")."""
        super().__init__()
        self.config = config
        self.precision = precision
        self.embeddings: torch.Tensor | None = None  # frozen (V+1, H) table, row 0 = padding
        self.table_rnorm: torch.Tensor | None = None
        self.table_bf16: torch.Tensor | None = None
        self.id2idx = None
        self._step = 0
        self._seed = int(seed)
        self.configure_model(device=device, seed=seed)
        if model is not None:
            self.load_encoder_state_dict(model)

    # ------------------------------------------------------------------ construction
    @staticmethod
    def resolve_config(config: ModelConfig) -> ModelConfig:
        """Fill the ``None`` fields of ``config`` in place, offline (``models.py:69-91`` does it by downloading
        ``pretrained_model_name``): a known model's published config first; otherwise ``hidden_size`` follows from the
        head count (head size 32). Anything still missing raises."""
        known = KNOWN_PRETRAINED.get(config.pretrained_model_name, {})
        for k in ("vocab_size", "max_seq_length", "hidden_size", "num_hidden_layers", "num_attention_heads",
                  "intermediate_size"):
            if getattr(config, k) is None and k in known:
                setattr(config, k, known[k])
        if config.hidden_size is None and config.num_attention_heads is not None:
            config.hidden_size = 32 * config.num_attention_heads
        if config.num_attention_heads is None and config.hidden_size is not None and config.hidden_size % 32 == 0:
            config.num_attention_heads = config.hidden_size // 32
        missing = [k for k in ("hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
                               "max_seq_length") if getattr(config, k) is None]
        if missing:
            raise ValueError(
                f"ModelConfig fields {missing} are None and {config.pretrained_model_name!r} is not in "
                f"params.KNOWN_PRETRAINED: the reference resolves them by downloading that model (models.py:69-91); "
                f"this build is offline -- set them explicitly"
            )
        return config

    def configure_model(self, device=None, seed: int = 0) -> None:
        c = self.resolve_config(self.config)
        if c.hidden_size not in (32 * c.num_attention_heads, 64 * c.num_attention_heads):
            raise ValueError(
                f"hidden_size / num_attention_heads must be 32 or 64 (got {c.hidden_size}/{c.num_attention_heads}): the "
                "gfx950 attention kernels are built for head size 32 (the reference's 384/12; the production kernels) "
                "and 64 (e.g. 384/6, 768/12; generic kernels)"
            )
        H, I, Lm, nL = c.hidden_size, c.intermediate_size, c.max_seq_length, c.num_hidden_layers
        names, shapes, offsets, total = flat_layout(H, I, Lm, nL)
        self._names, self._shapes, self._offsets = names, shapes, offsets
        g = torch.Generator().manual_seed(seed)
        flat = torch.zeros(total, dtype=torch.float32)
        for name, shape, off in zip(names, shapes, offsets):
            n = int(torch.tensor(shape).prod())
            if "LayerNorm.weight" in name:
                flat[off : off + n] = 1.0
            elif name.endswith(".weight"):  # HF BERT init: N(0, initializer_range); biases / LN bias = 0
                flat[off : off + n] = torch.randn(n, generator=g) * INITIALIZER_RANGE
        self.flat = torch.nn.Parameter(flat.to(device) if device is not None else flat)

    @property
    def device(self) -> torch.device:
        return self.flat.device

    @property
    def max_seq_length(self) -> int:
        return int(self.config.max_seq_length)

    def encoder_state_dict(self) -> dict[str, torch.Tensor]:
        """HF-BERT-keyed views into the flat buffer (the reference's checkpoint keys sit under
        ``model.model.0.auto_model.``: SURVEY section 5)."""
        out = {}
        for name, shape, off in zip(self._names, self._shapes, self._offsets):
            n = int(torch.tensor(shape).prod())
            out[name] = self.flat.detach()[off : off + n].view(shape)
        return out

    def load_encoder_state_dict(self, state: dict[str, torch.Tensor], strict: bool = True) -> None:
        views = self.encoder_state_dict()
        missing = [k for k in views if k not in state]
        if strict and missing:
            raise KeyError(f"missing encoder tensors: {missing[:4]}{'...' if len(missing) > 4 else ''}")
        with torch.no_grad():
            for k, v in views.items():
                if k in state:
                    v.copy_(torch.as_tensor(state[k]).to(v.device, torch.float32).reshape(v.shape))

    def grad_state_dict(self) -> dict[str, torch.Tensor]:
        if self.flat.grad is None:
            return {}
        out = {}
        for name, shape, off in zip(self._names, self._shapes, self._offsets):
            n = int(torch.tensor(shape).prod())
            out[name] = self.flat.grad[off : off + n].view(shape)
        return out

    def configure_embeddings(self, items_dataset) -> None:
        """Frozen item table with a zero padding row + ``id2idx`` (``models.py:234-259``).

        ``items_dataset``: a ``datasets.Dataset`` with ``embedding`` / ``item_id`` columns as in the
        reference, or any mapping with those two keys (arrays)."""
        if self.embeddings is None:
            if hasattr(items_dataset, "with_format"):
                weights = items_dataset.with_format("torch")["embedding"][:]
            else:
                weights = torch.as_tensor(items_dataset["embedding"])
            weights = weights.to(torch.float32)
            if weights.shape[1] != self.config.hidden_size:
                raise ValueError(
                    f"item embedding width {weights.shape[1]} != hidden_size {self.config.hidden_size}: "
                    "embeddings enter the encoder as inputs_embeds without projection (models.py:336-345)"
                )
            table = torch.cat([torch.zeros_like(weights[:1]), weights]).contiguous().to(self.device)
            self.set_table(table)
        if self.id2idx is None:
            if hasattr(items_dataset, "with_format"):
                ids = list(items_dataset.with_format("pandas")["item_id"].array)
            else:
                ids = list(items_dataset["item_id"])
            try:
                import pandas as pd

                self.id2idx = pd.Series(pd.RangeIndex(len(ids)) + 1, index=ids)
            except ImportError:  # pragma: no cover
                self.id2idx = {k: i + 1 for i, k in enumerate(ids)}

    def set_table(self, table: torch.Tensor) -> None:
        """Install a ready ``(V+1, H)`` table (row 0 = padding)."""
        if table.dim() != 2 or table.shape[1] != self.config.hidden_size:
            raise ValueError(
                f"item table of shape {tuple(table.shape)} does not match hidden_size {self.config.hidden_size}: "
                "embeddings enter the encoder as inputs_embeds without projection (models.py:336-345)")
        self.embeddings = table.contiguous()
        # frozen table => per-item inverse norms and the bf16 gather copy are computed once
        self.table_rnorm, self.table_bf16 = ops.table_prepare(self.embeddings) if table.is_cuda else (None, None)

    def load_table(self, path, *, column: str = "embedding", id_column: str = "item_id", has_padding_row=None) -> None:
        """Install the frozen item table from a FILE ("optionally accept any (V,H) table file";
        the reference only ever gets it through ``items_dataset``, ``models.py:234-259``).

        ``.npy`` (``allow_pickle=False``), ``.safetensors`` / ``.pt`` (``weights_only=True``): one 2-D float array, or a
        dict holding it under ``column`` (``"embeddings.weight"`` -- the reference module's own key -- is also looked up);
        ``.parquet``: the ``items.parquet`` layout of ``data.py:395-411`` -- a list column ``column`` plus ``id_column``,
        which also builds ``id2idx``. A ``(V, H)`` array gets the zero padding row prepended as ``configure_embeddings``
        does; ``has_padding_row=True`` (or an all-zero first row when left ``None``) says it is ``(V+1, H)`` already."""
        import numpy as np

        path = pathlib.Path(path)
        suffix = path.suffix.lower()
        ids = None
        if suffix == ".npy":
            arr = torch.from_numpy(np.load(path, allow_pickle=False))
        elif suffix == ".safetensors":
            from safetensors.torch import load_file

            d = load_file(str(path))
            arr = d[column] if column in d else d["embeddings.weight"] if "embeddings.weight" in d else next(iter(d.values()))
        elif suffix in (".pt", ".pth", ".bin"):
            d = torch.load(path, map_location="cpu", weights_only=True)
            if isinstance(d, dict):
                d = d[column] if column in d else d["embeddings.weight"] if "embeddings.weight" in d else next(iter(d.values()))
            arr = torch.as_tensor(d)
        elif suffix == ".parquet":
            import pyarrow.parquet as pq

            tbl = pq.read_table(str(path), columns=[c for c in (column, id_column) if c])
            arr = torch.from_numpy(np.asarray(tbl.column(column).to_pylist(), dtype=np.float32))
            if id_column in tbl.column_names:
                ids = tbl.column(id_column).to_pylist()
            has_padding_row = False if has_padding_row is None else has_padding_row
        else:
            raise ValueError(f"load_table: unsupported file type {suffix!r} (.npy, .safetensors, .pt, .parquet)")
        arr = arr.to(torch.float32)
        if arr.dim() != 2:
            raise ValueError(f"load_table: expected a 2-D (V, H) array, got shape {tuple(arr.shape)}")
        if has_padding_row is None:
            has_padding_row = bool((arr[0] == 0).all())
        if ids is not None:
            self.embeddings = None
            self.id2idx = None
            self.configure_embeddings({"embedding": arr[1:] if has_padding_row else arr, "item_id": ids})
            return
        table = arr if has_padding_row else torch.cat([torch.zeros_like(arr[:1]), arr])
        self.set_table(table.contiguous().to(self.device))

    # ------------------------------------------------------------------ compute
    def context(self):
        """This model's ``xfmr_context`` (side stream of the backward's weight-gradient GEMMs), created on first use."""
        ctx = getattr(self, "_context", None)
        if ctx is None and self.flat.is_cuda:
            ctx = self._context = ops.Context(self.device)
        return ctx

    def use_device_step(self, on: bool = True) -> None:
        """Dropout streams keyed by a step counter in HBM (``xfmr_encoder_cfg.step_device``) instead of the host-side
        step count: what a captured hipGraph of the training step needs (every replay a new mask). The counter is
        advanced by the optimizer (``FusedAdamW(step_device=...)``) as the last launch of a step."""
        self.step_device = torch.zeros(1, dtype=torch.int32, device=self.device) if on else None

    def _cfg(self, B: int, L: int, embed_event=None) -> N.EncoderCfg:
        return ops.make_encoder_cfg(**self._cfg_kwargs(B, L, embed_event))

    def supports_packed_rows(self, L: int) -> bool:
        """Whether the encoder can run the PACKED layout (xfmr_encoder_cfg.seq_offsets: the token axis holds each sequence's own
        rows only, no padding rows) for sequences of up to L rows: bf16 policy, head size 32, causal, L <= 256."""
        c = self.config
        return (self.precision == "bf16" and c.hidden_size == 32 * c.num_attention_heads and bool(c.is_decoder)
                and min(L, self.max_seq_length) <= 256 and self.flat.is_cuda and os.environ.get("XFMR_ACT_FP32", "0") in ("", "0")
                and os.environ.get("XFMR_PACKED", "1") != "0")

    def _cfg_kwargs(self, B: int, L: int, embed_event=None, packed=None) -> dict:
        """make_encoder_cfg's keyword arguments for this model, in plain Python (traced code builds the scalar arguments
        of torch.ops.xfmr.encoder from them; eager code the xfmr_encoder_cfg)."""
        c = self.config
        train = self.training
        step_dev = getattr(self, "step_device", None)
        if torch.compiler.is_compiling():
            # inside a traced region nothing may be created through ctypes: the side stream is used if it already exists
            # (model.context() before compiling), and the per-step part of the dropout seed has to live on the device
            ctx = getattr(self, "_context", None) if train else None
            if train and step_dev is None:
                raise RuntimeError("compiling a TRAINING step: call model.use_device_step(True) first (the dropout stream "
                                   "must be keyed by a device-side counter; a host-side step count would be baked in)")
        else:
            ctx = self.context() if train else None
        return dict(
            batch=B, seq_len=L, hidden=c.hidden_size, heads=c.num_attention_heads, inter=c.intermediate_size,
            layers=c.num_hidden_layers, max_pos=c.max_seq_length, precision=self.precision, ln_eps=LAYER_NORM_EPS,
            hidden_dropout=HIDDEN_DROPOUT_PROB if train else 0.0,
            attn_dropout=ATTENTION_PROBS_DROPOUT_PROB if train else 0.0,
            # with a device-side counter the host-side part of the seed stays fixed (a captured launch replays it)
            seed=(self._seed * 0x9E3779B97F4A7C15 + (0 if step_dev is not None else self._step)) & 0xFFFFFFFFFFFFFFFF,
            causal=bool(c.is_decoder),  # BertConfig(is_decoder=...) (models.py:355): False = key-padding mask only
            step_device=step_dev if train else None, embed_event=embed_event,
            context=ctx.handle if ctx is not None else None,
            grads_half_event=getattr(self, "grads_half_event", None),  # set by distributed.HalvedAllReduce
            extra_flags=getattr(self, "enc_flags", 0),  # e.g. ENC_DW_SIDE_ANY inside a captured step (GraphedStep)
            profile=getattr(self, "enc_profile", None) if train else None,  # (kernel, layer, ev0, ev1): bench.py
            seq_offsets=packed["seq_offsets"] if packed else None, row_pos=packed["row_pos"] if packed else None,
        )

    def _encode_tokens(self, item_idx=None, item_embeds=None, embed_event=None, packed=None):
        """tok, key_mask of the history. ``packed`` (ops.pack_rows's dict + "batch" / "seq_len"): the packed layout -- tok is
        (rows, H), key_mask (rows,): only each sequence's own rows, in order."""
        assert self.embeddings is not None, "call configure_embeddings() first"
        if packed is not None:
            if self.training and not torch.compiler.is_compiling():
                self._step += 1
            kw = self._cfg_kwargs(int(packed["batch"]), int(packed["seq_len"]), embed_event, packed)
            return ops.encoder(self.flat, packed["hist"], self.embeddings, kw)
        if item_embeds is not None:
            # the gather kernel reads rows by index: use the given embeddings as the table
            x = item_embeds[:, -self.max_seq_length :, :].to(self.device, torch.float32).contiguous()
            B, L, H = x.shape
            table = x.view(B * L, H)
            idx = torch.arange(B * L, device=self.device, dtype=torch.int64).view(B, L)
        elif item_idx is not None:
            idx = item_idx[:, -self.max_seq_length :].to(self.device, torch.int64).contiguous()
            table = self.embeddings
            B, L = idx.shape
        else:
            msg = "either `item_idx` or `item_embeds` must be provided"
            raise ValueError(msg)
        if self.training and not torch.compiler.is_compiling():  # (traced training steps key dropout on the device)
            self._step += 1
        # torch.ops.xfmr.encoder when traced (custom_ops.py), the autograd.Function over the same C-ABI calls in eager mode
        tok, key_mask = ops.encoder(self.flat, idx, table, self._cfg_kwargs(B, L, embed_event))
        return tok, key_mask

    def forward(self, item_idx=None, *, item_embeds=None) -> dict[str, torch.Tensor]:
        """``models.py:306-345``: returns token_embeddings (B,L,H), sentence_embedding (B,H), attention_mask."""
        tok, key_mask = self._encode_tokens(item_idx, item_embeds)
        with torch.no_grad():  # sentence_embedding is not on the training path (models.py:143-147)
            sent = ops.pool(tok.detach(), key_mask, self.config.pooling_mode)
            if self.config.is_normalized:
                sent = ops.l2_normalize(sent)
        return {"token_embeddings": tok, "sentence_embedding": sent, "attention_mask": key_mask.long()}

    def encode(self, item_ids: list[str]) -> torch.Tensor:
        """Pooled embedding of a list of item ids; unknown ids are dropped (``models.py:347-364``)."""
        assert self.id2idx is not None
        known = [i for i in item_ids if i in self.id2idx.index] if hasattr(self.id2idx, "index") else [
            i for i in item_ids if i in self.id2idx
        ]
        vals = self.id2idx[known].to_numpy() if hasattr(self.id2idx, "index") else [self.id2idx[i] for i in known]
        item_idx = torch.as_tensor(vals, device=self.device, dtype=torch.int64)
        return self(item_idx[None, :])["sentence_embedding"][0]

    def compute_embeds(self, history_item_idx, pos_item_idx, neg_item_idx) -> dict:
        """``models.py:366-419``. ``query_embed`` is the compacted ``(Np, H)`` tensor (this needs the row count
        on the host, as the reference's boolean indexing does); ``candidate_embed`` is structured."""
        tok, key_mask = self._encode_tokens(history_item_idx)
        am = key_mask.bool()
        pos = pos_item_idx.to(self.device)[am]
        neg = neg_item_idx.to(self.device)[am]
        keep = pos != 0
        query = tok[am][keep]
        if self.config.is_normalized:  # models.py:393-394 (normalising before or after the row selection is the same)
            query = ops.l2_normalize(query)
        cand = SharedNegatives(self.embeddings, self.table_rnorm, pos[keep], neg, table_bf16=self.table_bf16)
        return {"query_embed": query, "candidate_embed": cand, "attention_mask": am, "positive_mask": keep}

    # ------------------------------------------------------------------ persistence
    def save(self, path: str) -> None:
        """``models.py:261-269``: the SentenceTransformer directory layout (see write_sentence_transformer_dir)."""
        write_sentence_transformer_dir(
            path, self.config, {k: v.contiguous().cpu() for k, v in self.encoder_state_dict().items()}
        )

    @classmethod
    def load(cls, path: str, device=None, precision: str = "bf16") -> "RecommenderModel":
        """``models.py:271-304``: rebuild the config from the saved directory (HF ``config.json``, ``1_Pooling``,
        ``modules.json``) and load the encoder tensors."""
        config, state = read_sentence_transformer_dir(path)
        return cls(config, device=device, model=state, precision=precision)


POOLING_FLAGS = {  # sentence-transformers Pooling config keys for the modes ModelConfig allows
    "cls": "pooling_mode_cls_token", "mean": "pooling_mode_mean_tokens", "max": "pooling_mode_max_tokens",
    "lasttoken": "pooling_mode_lasttoken",
}


def write_sentence_transformer_dir(path, config: ModelConfig, encoder_state: dict) -> None:
    """The directory ``SentenceTransformer.save`` produces for the reference's model (``models.py:104-149, 261-269``;
    sentence-transformers 5.7 layout): ``modules.json`` (Transformer, Pooling[, Normalize]), the HF BertModel at the
    root (``config.json`` + ``model.safetensors`` with ``BertModel`` keys), ``sentence_bert_config.json``,
    ``1_Pooling/config.json``. ``word_embeddings`` and ``pooler.*`` never train on this path: they are
    written with their (deterministic) HF init so that ``BertModel.from_pretrained`` loads the directory strictly.
    Tokenizer files belong to ``config.pretrained_model_name`` and cannot be produced offline: copy them next to
    these files before loading with sentence-transformers (the training path feeds ``inputs_embeds`` only)."""
    from safetensors.torch import save_file

    p = pathlib.Path(path)
    p.mkdir(parents=True, exist_ok=True)
    c = config
    H = c.hidden_size
    state = {k: torch.as_tensor(v).detach().to(torch.float32).contiguous().cpu() for k, v in encoder_state.items()}
    g = torch.Generator().manual_seed(0)
    state.setdefault("embeddings.word_embeddings.weight", torch.randn(c.vocab_size or 1, H, generator=g) * 0.02)
    state.setdefault("pooler.dense.weight", torch.randn(H, H, generator=g) * 0.02)
    state.setdefault("pooler.dense.bias", torch.zeros(H))
    save_file(state, str(p / "model.safetensors"), metadata={"format": "pt"})
    (p / "config.json").write_text(json.dumps({
        "architectures": ["BertModel"], "model_type": "bert", "is_decoder": c.is_decoder,
        "vocab_size": c.vocab_size or 1, "hidden_size": H, "num_hidden_layers": c.num_hidden_layers,
        "num_attention_heads": c.num_attention_heads, "intermediate_size": c.intermediate_size,
        "max_position_embeddings": c.max_seq_length, "type_vocab_size": 2, "hidden_act": "gelu",
        "hidden_dropout_prob": 0.1, "attention_probs_dropout_prob": 0.1, "layer_norm_eps": 1e-12,
        "initializer_range": 0.02, "pad_token_id": 0, "position_embedding_type": "absolute", "use_cache": True,
        "torch_dtype": "float32",
        # build-specific extras (ignored by HF): lets RecommenderModel.load round-trip without sentence-transformers
        "xfmr_pooling_mode": c.pooling_mode, "xfmr_is_normalized": c.is_normalized,
        "xfmr_pretrained_model_name": c.pretrained_model_name,
    }, indent=2))
    modules = [
        {"idx": 0, "name": "0", "path": "", "type": "sentence_transformers.models.Transformer"},
        {"idx": 1, "name": "1", "path": "1_Pooling", "type": "sentence_transformers.models.Pooling"},
    ]
    if c.is_normalized:
        modules.append({"idx": 2, "name": "2", "path": "2_Normalize", "type": "sentence_transformers.models.Normalize"})
        (p / "2_Normalize").mkdir(exist_ok=True)
    (p / "modules.json").write_text(json.dumps(modules, indent=2))
    (p / "sentence_bert_config.json").write_text(json.dumps({"max_seq_length": c.max_seq_length, "do_lower_case": False}, indent=2))
    (p / "config_sentence_transformers.json").write_text(json.dumps(
        {"__version__": {"sentence_transformers": "5.7.0"}, "prompts": {}, "default_prompt_name": None,
         "similarity_fn_name": "cosine"}, indent=2))
    (p / "1_Pooling").mkdir(exist_ok=True)
    pool = {"word_embedding_dimension": H, "include_prompt": True}
    pool |= {flag: mode == c.pooling_mode for mode, flag in POOLING_FLAGS.items()}
    pool |= {"pooling_mode_mean_sqrt_len_tokens": False, "pooling_mode_weightedmean_tokens": False}
    (p / "1_Pooling" / "config.json").write_text(json.dumps(pool, indent=2))


def read_sentence_transformer_dir(path):
    """(ModelConfig, HF-keyed encoder state) from a directory written by :func:`write_sentence_transformer_dir` or by
    ``SentenceTransformer.save`` of the reference's model."""
    from safetensors.torch import load_file

    p = pathlib.Path(path)
    d = json.loads((p / "config.json").read_text())
    pooling_mode = d.get("xfmr_pooling_mode", d.get("pooling_mode", "mean"))
    pc = p / "1_Pooling" / "config.json"
    if pc.exists():
        flags = json.loads(pc.read_text())
        for mode, flag in POOLING_FLAGS.items():
            if flags.get(flag):
                pooling_mode = mode
    is_normalized = d.get("xfmr_is_normalized", d.get("is_normalized", False))
    mj = p / "modules.json"
    if mj.exists():
        is_normalized = any(m.get("type", "").endswith("Normalize") for m in json.loads(mj.read_text()))
    config = ModelConfig(
        vocab_size=d.get("vocab_size", 1), hidden_size=d["hidden_size"], num_hidden_layers=d["num_hidden_layers"],
        num_attention_heads=d["num_attention_heads"], intermediate_size=d["intermediate_size"],
        max_seq_length=d["max_position_embeddings"], is_decoder=d.get("is_decoder", True),
        pretrained_model_name=d.get("xfmr_pretrained_model_name", d.get("pretrained_model_name", PRETRAINED_MODEL_NAME)),
        pooling_mode=pooling_mode, is_normalized=is_normalized,
    )
    state = load_file(str(p / "model.safetensors"))
    state = {(k[len("bert."):] if k.startswith("bert.") else k): v for k, v in state.items()}
    return config, state
