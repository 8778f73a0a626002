#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Run in the build container only (it reads /root/reference, which never
travels to the GPU box):

    python oracle/make_golden.py

What is executed from the reference:
  * ``xfmr_rec/losses.py`` is imported unmodified (it needs only
    abc/typing/pydantic/torch) -> G1 loss vectors, and the loss stage of G3.
  * the encoder is constructed exactly as ``xfmr_rec/models.py:93-102`` does:
    ``BertModel(BertConfig(vocab_size, hidden_size, num_hidden_layers,
    num_attention_heads, intermediate_size, max_position_embeddings,
    is_decoder=True))`` from the ``transformers`` package in this container
    (version recorded in every fixture) -> G2, and the encoder stage of G3;
    the same with ``is_decoder=False`` (``ModelConfig.is_decoder``,
    models.py:50) -> G5.
  * ``xfmr_rec/models.py`` / ``trainer.py`` are NOT importable here (they need
    loguru / sentence_transformers / lightning, absent offline). Their glue
    (``models.py:333-345, 366-419``, ``trainer.py:213-264``) is restated in
    ``_ref_compute_embeds`` / ``_ref_compute_losses`` below, around the real
    HF encoder and the real reference loss classes.

Fixtures are plain ``.npz`` (no pickles): inputs + expected outputs only.
"""

from __future__ import annotations

import itertools
import json
import pathlib
import sys

import numpy as np
import torch

REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parents[1] / "tests" / "golden"
sys.path.insert(0, str(REF))

import transformers  # noqa: E402
import xfmr_rec.losses as ref_losses  # noqa: E402  (the reference, unmodified)
from transformers.models.bert import BertConfig, BertModel  # noqa: E402

VERSIONS = json.dumps({"torch": torch.__version__, "transformers": transformers.__version__})


def _np(t):
    return t.detach().cpu().numpy().copy()  # copy: AdamW updates parameters in place


# --------------------------------------------------------------------------- G1
def gen_losses():
    """7 heads x config variants on small random (q, cand); loss, logits, mask, dL/dq, dL/dcand."""
    variants = [
        dict(),
        dict(mask_false_negatives=False),
        dict(num_hard_negatives=3),
        dict(scale=20.0),
        dict(margin=0.0),
        dict(target_position="diagonal"),
        dict(target_position=None),  # explicit target tensor
        dict(mask_false_negatives=False, num_hard_negatives=2, scale=5.0, margin=0.25),
    ]
    shapes = [(5, 8, 8), (19, 23, 16)]
    store: dict[str, np.ndarray] = {"versions": np.array(VERSIONS)}
    index = []
    case = 0
    for seed, (n, c, h) in itertools.product(range(1), shapes):
        g = torch.Generator().manual_seed(100 + seed)
        q0 = torch.randn(n, h, generator=g)
        c0 = torch.randn(n, c, h, generator=g)
        # plant exact duplicates of the positive among the candidates (false negatives)
        c0[1, 3] = c0[1, 0]
        c0[n - 1, c - 1] = c0[n - 1, 0]
        tgt = torch.randint(0, c, (n,), generator=g)
        for vi, var in enumerate(variants):
            cfg = ref_losses.LossConfig(**var)
            target = tgt if cfg.target_position is None else None
            if cfg.target_position == "diagonal" and c < n:
                continue
            for cls in ref_losses.LOSS_CLASSES:
                q = q0.clone().requires_grad_(True)
                cand = c0.clone().requires_grad_(True)
                fn = cls(cfg)
                logits = fn.compute_logits(q, cand)
                t2 = fn.check_target(logits, target)
                mask = fn.mine_hard_negatives(logits, fn.mask_false_negatives(logits, t2))
                loss = fn(q, cand, target)
                loss.backward()
                key = f"c{case:04d}"
                index.append(
                    dict(key=key, kind=cls.__name__, seed=seed, shape=[n, c, h], variant=vi,
                         cfg=cfg.model_dump(), has_target=target is not None)
                )
                store[f"{key}/loss"] = _np(loss)
                store[f"{key}/logits"] = _np(logits)
                store[f"{key}/mask"] = _np(mask)
                store[f"{key}/dq"] = _np(q.grad)
                store[f"{key}/dcand_absmax"] = _np(cand.grad.abs().amax())
                store[f"{key}/dcand_sum"] = _np(cand.grad.sum((0, 1)))
                case += 1
            stats = ref_losses.LogitsStatistics(cfg)(q0, c0, target)
            store[f"stats/s{seed}_n{n}_v{vi}"] = np.array(json.dumps(stats))
        store[f"in/s{seed}_n{n}/q"] = _np(q0)
        store[f"in/s{seed}_n{n}/cand"] = _np(c0)
        store[f"in/s{seed}_n{n}/target"] = _np(tgt)
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(OUT / "g1_losses.npz", **store)
    print("g1_losses:", case, "cases")


# --------------------------------------------------------------------------- G2
def _make_bert(H, nL, A, I, Lmax, attn_impl, seed=0, is_decoder=True):
    torch.manual_seed(seed)
    cfg = BertConfig(
        vocab_size=1, hidden_size=H, num_hidden_layers=nL, num_attention_heads=A,
        intermediate_size=I, max_position_embeddings=Lmax, is_decoder=is_decoder,
    )
    cfg._attn_implementation = attn_impl
    m = BertModel(cfg)
    # HF init leaves LN = (1, 0) and biases = 0; perturb them so the golden exercises them
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, p in m.named_parameters():
            if name.endswith("bias") or "LayerNorm.weight" in name:
                p.add_(0.1 * torch.randn(p.shape, generator=g))
    return m.eval()


def _ragged_inputs(B, L, H, seed, lengths):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, L, H, generator=g)
    x = x / x.norm(dim=-1, keepdim=True)
    for b, n in enumerate(lengths):
        x[b, n:] = 0.0
    return x


def gen_encoder(is_decoder=True, out_name="g2_encoder.npz"):
    store: dict[str, np.ndarray] = {"versions": np.array(VERSIONS)}
    cases = {
        # head size 32 throughout (the reference's 384/12; the HIP attention kernels are built for it)
        "a": dict(B=3, L=12, H=32, nL=2, A=1, I=64, lengths=[12, 7, 1]),
        "b": dict(B=2, L=20, H=64, nL=1, A=2, I=64, lengths=[20, 13]),
        "c": dict(B=2, L=40, H=32, nL=2, A=1, I=32, lengths=[33, 40]),
    }
    if not is_decoder:  # G5: ModelConfig.is_decoder=False (models.py:50,355) -- bidirectional attention
        cases["d"] = dict(B=2, L=150, H=64, nL=1, A=2, I=64, lengths=[150, 41])  # two 128-row blocks
    for name, c in cases.items():
        x = _ragged_inputs(c["B"], c["L"], c["H"], 7, c["lengths"])
        mask = (x != 0).any(-1).long()
        state = None
        for impl in ("eager", "sdpa"):
            m = _make_bert(c["H"], c["nL"], c["A"], c["I"], c["L"], impl, is_decoder=is_decoder)
            if state is None:
                state = {k: v.clone() for k, v in m.state_dict().items()}
            m.zero_grad()
            xin = x.clone().requires_grad_(True)
            out = m(inputs_embeds=xin, attention_mask=mask).last_hidden_state
            # loss touches only valid tokens, like the training path (models.py:392)
            w = torch.linspace(0.5, 1.5, c["H"])
            loss = ((out * w) ** 2 * mask[..., None]).sum()
            loss.backward()
            store[f"{name}/{impl}/last_hidden_state"] = _np(out)
            store[f"{name}/{impl}/loss"] = _np(loss)
            store[f"{name}/{impl}/dx"] = _np(xin.grad)
            if impl == "eager":
                for k, p in m.named_parameters():
                    if p.grad is not None:
                        store[f"{name}/{impl}/grad/{k}"] = _np(p.grad)
        for k, v in state.items():
            if "word_embeddings" in k or "pooler" in k:
                continue
            store[f"{name}/param/{k}"] = _np(v)
        store[f"{name}/x"] = _np(x)
        store[f"{name}/mask"] = _np(mask)
        store[f"{name}/cfg"] = np.array(json.dumps({k: v for k, v in c.items()}))
    store["is_decoder"] = np.array(is_decoder)
    np.savez_compressed(OUT / out_name, **store)
    print(out_name, list(cases))


# --------------------------------------------------------------------------- G3
def _ref_compute_embeds(bert, table, hist, pos, neg, max_seq_length, is_normalized=False):
    """models.py:333-345 + 366-419, restated around the real HF encoder."""
    x = torch.nn.functional.embedding(hist[:, -max_seq_length:], table)
    attention_mask = (x != 0).any(-1).long()
    tok = bert(inputs_embeds=x, attention_mask=attention_mask).last_hidden_state
    am = attention_mask.bool()
    query = tok[am]
    if is_normalized:
        query = torch.nn.functional.normalize(query, dim=-1)
    pos_i = pos[am]
    pos_e = torch.nn.functional.embedding(pos_i, table)[:, None, :]
    neg_e = torch.nn.functional.embedding(neg[am], table)
    neg_e = neg_e[None, :, :].expand(pos_e.size(0), -1, -1)
    cand = torch.cat([pos_e, neg_e], dim=1)
    keep = pos_i != 0
    return query[keep], cand[keep], am, keep, tok


def gen_step():
    store: dict[str, np.ndarray] = {"versions": np.array(VERSIONS)}
    V, H, nL, A, I, L, B = 50, 64, 1, 2, 32, 10, 4
    g = torch.Generator().manual_seed(3)
    items = torch.randn(V, H, generator=g)
    items = items / items.norm(dim=-1, keepdim=True)
    table = torch.cat([torch.zeros(1, H), items])
    lengths = [10, 6, 3, 10]
    hist = torch.zeros(B, L, dtype=torch.long)
    pos = torch.zeros(B, L, dtype=torch.long)
    neg = torch.zeros(B, L, dtype=torch.long)
    for b, n in enumerate(lengths):
        hist[b, :n] = torch.randint(1, V + 1, (n,), generator=g)
        pos[b, :n] = torch.randint(1, V + 1, (n,), generator=g)
        neg[b, :n] = torch.randint(1, V + 1, (n,), generator=g)
    pos[1, 5] = 0  # a valid history position whose positive is padding (models.py:413)
    neg[0, 2] = pos[0, 4]  # a sampled negative that equals another row's positive
    neg[0, 4] = pos[0, 4]  # ... and one that equals its own row's positive (false negative)

    for train_loss in ("InfoNCELoss", "PairwiseLogisticLoss", "AlignmentContrastiveLoss"):
        bert = _make_bert(H, nL, A, I, L, "eager", seed=11).train(False)
        params = [p for n, p in bert.named_parameters()]
        cfg = ref_losses.LossConfig()
        q, cand, am, keep, tok = _ref_compute_embeds(bert, table, hist, pos, neg, L)
        if train_loss == "InfoNCELoss":
            store["tok"] = _np(tok)
            store["query_embed"] = _np(q)
            store["attention_mask"] = _np(am)
            store["positive_mask"] = _np(keep)
            store["stats"] = np.array(json.dumps(ref_losses.LogitsStatistics(cfg)(q, cand)))
            for cls in ref_losses.LOSS_CLASSES:
                store[f"loss/{cls.__name__}"] = _np(cls(cfg)(q, cand))
            for k, v in bert.state_dict().items():
                if "word_embeddings" not in k and "pooler" not in k:
                    store[f"param0/{k}"] = _np(v)
        # three AdamW steps (trainer.py:327-332), dropout off
        opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.01)
        loss_fn = dict((c.__name__, c) for c in ref_losses.LOSS_CLASSES)[train_loss](cfg)
        for step in range(3):
            opt.zero_grad(set_to_none=True)
            q, cand, *_ = _ref_compute_embeds(bert, table, hist, pos, neg, L)
            loss = loss_fn(q, cand)
            loss.backward()
            if step == 0 and train_loss != "AlignmentContrastiveLoss":
                for k, p in bert.named_parameters():
                    if p.grad is not None:
                        store[f"{train_loss}/grad0/{k}"] = _np(p.grad)
            store[f"{train_loss}/loss_step{step}"] = _np(loss)
            opt.step()
            if step == 2 or (step == 0 and train_loss == "InfoNCELoss"):
                for k, p in bert.named_parameters():
                    if "word_embeddings" not in k and "pooler" not in k:
                        store[f"{train_loss}/param_after{step + 1}/{k}"] = _np(p)
    store["table"] = _np(table)
    store["hist"] = _np(hist)
    store["pos"] = _np(pos)
    store["neg"] = _np(neg)
    store["cfg"] = np.array(json.dumps(dict(V=V, H=H, nL=nL, A=A, I=I, L=L, B=B)))
    np.savez_compressed(OUT / "g3_step.npz", **store)
    print("g3_step done")


# --------------------------------------------------------------------------- G4
def gen_shared_negatives():
    """The training-path candidate structure (models.py:398-416) through the reference loss classes:
    cand = cat([E[pos][:, None], E[neg][None].expand(Np, -1, -1)], 1), positive at column 0."""
    store: dict[str, np.ndarray] = {"versions": np.array(VERSIONS)}
    V, H, Np, Nn = 200, 64, 21, 33
    g = torch.Generator().manual_seed(5)
    items = torch.randn(V, H, generator=g)
    items = items / items.norm(dim=-1, keepdim=True)
    table = torch.cat([torch.zeros(1, H), items])
    q0 = torch.randn(Np, H, generator=g) * 1.5
    # positives from ids 1..100, negatives from 101..200 (+ two padding-row negatives): NO row has a negative
    # that is its own positive, so no logit tie exists and the reference's result does not depend on the
    # rounding of its bmm (see the 'ties' block below for what happens when one does).
    pos = torch.randint(1, 101, (Np,), generator=g)
    neg = torch.randint(101, V + 1, (Nn,), generator=g)
    neg[10] = 0
    neg[20] = 0
    variants = [
        dict(),
        dict(mask_false_negatives=False),
        dict(scale=20.0),
        dict(margin=0.0),
        dict(mask_false_negatives=False, scale=5.0, margin=0.25),
    ]
    index = []
    for vi, var in enumerate(variants):
        cfg = ref_losses.LossConfig(**var)
        for cls in ref_losses.LOSS_CLASSES:
            q = q0.clone().requires_grad_(True)
            cand = torch.cat([table[pos][:, None, :], table[neg][None, :, :].expand(Np, -1, -1)], dim=1)
            loss = cls(cfg)(q, cand)
            loss.backward()
            key = f"v{vi}/{cls.__name__}"
            store[f"{key}/loss"] = _np(loss)
            store[f"{key}/dq"] = _np(q.grad)
            index.append(dict(key=key, kind=cls.__name__, cfg=cfg.model_dump()))
        cand = torch.cat([table[pos][:, None, :], table[neg][None, :, :].expand(Np, -1, -1)], dim=1)
        store[f"v{vi}/stats"] = np.array(json.dumps(ref_losses.LogitsStatistics(cfg)(q0, cand)))
    # full-catalogue form: every table row is a column, target = the positive's row
    cfg = ref_losses.LossConfig(target_position=None, mask_false_negatives=False)
    for cls in ref_losses.LOSS_CLASSES:
        q = q0.clone().requires_grad_(True)
        loss = cls(cfg)(q, table[None].expand(Np, -1, -1), pos)
        loss.backward()
        store[f"catalog/{cls.__name__}/loss"] = _np(loss)
        store[f"catalog/{cls.__name__}/dq"] = _np(q.grad)
    # exact ties: two sampled negatives ARE some row's positive item. Mathematically their logit equals the
    # positive's and `logits < pos_logit` drops them; what the reference actually does depends on how its bmm
    # rounds column 0 vs column j, so its mask bits at the tie entries are recorded next to its loss.
    neg_t = neg.clone()
    neg_t[3] = pos[0]
    neg_t[7] = pos[5]
    neg_t[8] = pos[5]
    cfg = ref_losses.LossConfig()
    cand = torch.cat([table[pos][:, None, :], table[neg_t][None, :, :].expand(Np, -1, -1)], dim=1)
    tie_entries = torch.cat([torch.zeros(Np, 1, dtype=torch.bool), neg_t[None, :] == pos[:, None]], dim=1)
    for cls in ref_losses.LOSS_CLASSES:
        fn = cls(cfg)
        logits = fn.compute_logits(q0, cand)
        mask = fn.mask_false_negatives(logits, fn.check_target(logits, None))
        store[f"ties/{cls.__name__}/loss"] = _np(fn(q0, cand))
        store[f"ties/{cls.__name__}/tie_counted_as_negative"] = _np(mask[tie_entries])
    store["ties/neg"] = _np(neg_t)
    store["table"] = _np(table)
    store["q"] = _np(q0)
    store["pos"] = _np(pos)
    store["neg"] = _np(neg)
    store["index"] = np.array(json.dumps(index))
    np.savez_compressed(OUT / "g4_shared_negatives.npz", **store)
    print("g4_shared_negatives:", len(index), "cases + catalogue")


if __name__ == "__main__":
    OUT.mkdir(parents=True, exist_ok=True)
    torch.set_num_threads(4)
    only = sys.argv[1:]  # e.g. `python oracle/make_golden.py g5` regenerates one file

    def want(tag):
        return not only or tag in only

    if want("g1"):
        gen_losses()
    if want("g2"):
        gen_encoder()
    if want("g3"):
        gen_step()
    if want("g4"):
        gen_shared_negatives()
    if want("g5"):
        gen_encoder(is_decoder=False, out_name="g5_encoder_bidirectional.npz")
    for f in sorted(OUT.glob("*.npz")):
        print(f.name, f.stat().st_size // 1024, "KiB")
