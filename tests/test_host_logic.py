"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol the header declares, the
flat-parameter layout agrees between Python and C, host-side validation is loud, and the data-parallel
glue (sharding + single flat all-reduce) is equivalent to averaging independent per-shard gradients
(SURVEY section 8e), exercised with gloo at world_size 2."""

import ctypes as C
import os
import pathlib
import re
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = pathlib.Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def native():
    from xfmr_rec_amd import _native as N

    N.load()
    return N


def test_library_exports_every_declared_symbol(native):
    header = (ROOT / "include" / "xfmr_hip.h").read_text()
    declared = set(re.findall(r"\b(xfmr_[a-z0-9_]+)\s*\(", header))
    assert len(declared) >= 25
    lib = C.CDLL(str(native.LIB_PATH))
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f"declared in include/xfmr_hip.h but not exported: {missing}"
    assert declared == set(native.EXPORTED_SYMBOLS), declared ^ set(native.EXPORTED_SYMBOLS)
    assert native.load().xfmr_abi_version() == native.ABI_VERSION == 3
    assert native.load().xfmr_strerror(-2).decode().startswith("shape not supported")


@pytest.mark.parametrize("H,A,I,nL,Lmax", [(64, 2, 256, 2, 50), (128, 4, 512, 4, 200), (256, 8, 1024, 6, 200)])
def test_flat_layout_matches_c_abi(native, H, A, I, nL, Lmax):
    from xfmr_rec_amd import models, ops

    names, shapes, offsets, total = models.flat_layout(H, I, Lmax, nL)
    cfg = ops.make_encoder_cfg(batch=2, seq_len=Lmax, hidden=H, heads=A, inter=I, layers=nL, max_pos=Lmax,
                               precision="bf16")
    lib = native.load()
    assert lib.xfmr_param_count(C.byref(cfg)) == total
    buf = (C.c_int64 * (4 + 16 * nL))()
    n = lib.xfmr_param_offsets(C.byref(cfg), buf, len(buf))
    assert n == len(names) == len(offsets)
    assert list(buf) == offsets
    assert all(o % 4 == 0 for o in offsets)  # every tensor starts 16-byte aligned inside the flat buffer
    # SURVEY section 8e census of trainable fp32 elements (excludes word_embeddings / pooler)
    census = {(64, 256, 2, 50): 103_424, (128, 512, 4, 200): 819_200}
    if (H, I, nL, Lmax) in census:
        assert total == census[(H, I, nL, Lmax)]
    assert lib.xfmr_encoder_workspace_bytes(C.byref(cfg)) > 0


def test_unsupported_shapes_are_rejected_on_the_host(native):
    from xfmr_rec_amd import ops

    lib = native.load()
    bad = ops.make_encoder_cfg(batch=2, seq_len=8, hidden=48, heads=1, inter=64, layers=1, max_pos=8, precision="bf16")
    assert lib.xfmr_encoder_workspace_bytes(C.byref(bad)) == 0  # head size 48
    ok64 = ops.make_encoder_cfg(batch=2, seq_len=8, hidden=128, heads=2, inter=64, layers=1, max_pos=8, precision="bf16")
    assert lib.xfmr_encoder_workspace_bytes(C.byref(ok64)) > 0  # head size 64: the generic attention kernels
    long = ops.make_encoder_cfg(batch=2, seq_len=16, hidden=64, heads=2, inter=64, layers=1, max_pos=8, precision="bf16")
    assert lib.xfmr_encoder_workspace_bytes(C.byref(long)) == 0  # seq_len > max_pos
    assert lib.xfmr_sampled_loss_workspace(1000, 128, 500) > 0
    assert lib.xfmr_linear_bwd_dw_workspace(25600, 128, 128) >= 128 * 128 * 4


def test_weight_gradient_slab_room_covers_every_smaller_token_count(native):
    """The packed layout (ABI 3) carves the slab buffers for batch x seq_len tokens and launches with the real row count;
    the slab plan is not monotone in the token count (4 095 tokens: 32 slabs, 4 096: 16) -- round 4's e2e NaN. The size
    query must hold for every token count up to the one asked for."""
    lib = native.load()
    for n, k in ((192, 64), (128, 64), (64, 128), (384, 128), (512, 128), (768, 256), (1024, 256)):
        sizes = [lib.xfmr_linear_bwd_dw_workspace(m, n, k) for m in range(1, 9000, 37)]
        assert all(a <= b for a, b in zip(sizes, sizes[1:])), (n, k)
    assert lib.xfmr_linear_bwd_dw_workspace(4096, 192, 64) >= 32 * 192 * 64 * 4


def test_comm_entry_points_validate_on_the_host(native):
    """xfmr_comm_* / xfmr_allreduce_flat (K19): argument checks return codes, never touch a device; RCCL is resolved at run
    time (the library does not link against it)."""
    import subprocess

    lib = native.load()
    assert lib.xfmr_allreduce_flat(None, None, 0, None) == -1
    assert lib.xfmr_comm_create(None, None, 1, 0) == -1
    assert lib.xfmr_comm_destroy(None) == -1
    assert lib.xfmr_comm_unique_id(None) == -1
    assert b"RCCL" in lib.xfmr_strerror(native.ECOMM)
    needed = subprocess.run(["readelf", "-d", str(native.LIB_PATH)], capture_output=True, text=True).stdout
    assert "rccl" not in needed.lower()


def test_no_cpu_fallback():
    from xfmr_rec_amd import ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.layernorm_fwd(torch.zeros(4, 64), torch.ones(64), torch.zeros(64))


def test_config_surface_matches_reference_fields():
    """Field names and defaults are the user-visible contract (models.py:22-48, losses.py:11-30, trainer.py:98-102)."""
    import xfmr_rec_amd as X

    m = X.ModelConfig()
    assert (m.vocab_size, m.hidden_size, m.num_hidden_layers, m.num_attention_heads, m.intermediate_size,
            m.max_seq_length, m.is_decoder, m.pooling_mode, m.is_normalized) == (1, None, 1, 12, 48, 32, True, "mean", False)
    assert m.pretrained_model_name == "sentence-transformers/all-MiniLM-L6-v2"
    lc = X.LossConfig()
    assert (lc.target_position, lc.mask_false_negatives, lc.num_hard_negatives, lc.scale, lc.margin) == ("first", True, 0, 1.0, 0.5)
    t = X.LightningConfig(hidden_size=64, num_attention_heads=2)
    assert (t.train_loss, t.learning_rate, t.weight_decay, t.top_k) == ("InfoNCELoss", 0.001, 0.01, 20)
    assert [c.__name__ for c in X.LOSS_CLASSES] == [
        "AlignmentLoss", "AlignmentContrastiveLoss", "ContrastiveLoss", "InfoNCELoss", "NCELoss",
        "PairwiseHingeLoss", "PairwiseLogisticLoss"]
    with pytest.raises(ValueError, match="offline"):  # an unknown model name and nothing to derive the sizes from
        X.RecommenderModel(X.ModelConfig(pretrained_model_name="someone/unknown-model", num_attention_heads=None))
    with pytest.raises(ValueError, match="must be 32 or 64"):
        X.RecommenderModel(X.ModelConfig(hidden_size=48, num_attention_heads=1))
    assert X.RecommenderModel(X.ModelConfig(hidden_size=128, num_attention_heads=2)).config.hidden_size == 128  # head size 64


# The `model.config` block of the reference's config.yaml, verbatim (config.yaml:46-79): `hidden_size: null` is what
# `uv run train` starts from.
REFERENCE_CONFIG_YAML_MODEL_BLOCK = """
    vocab_size: 1
    hidden_size: null
    num_hidden_layers: 1
    num_attention_heads: 12
    intermediate_size: 48
    max_seq_length: 32
    is_decoder: true
    pretrained_model_name: sentence-transformers/all-MiniLM-L6-v2
    pooling_mode: mean
    is_normalized: false
    target_position: first
    mask_false_negatives: true
    num_hard_negatives: 0
    scale: 1.0
    margin: 0.5
    train_loss: InfoNCELoss
    learning_rate: 0.001
    weight_decay: 0.01
    items_config:
      id_col: item_id
      embedding_col: embedding
      lancedb_path: lance_db
      table_name: items
      text_col: item_text
      index_metric: cosine
    users_config:
      id_col: user_id
      embedding_col: null
      lancedb_path: lance_db
      table_name: users
      text_col: user_text
      index_metric: cosine
    top_k: 20
"""


def test_module_builds_from_the_reference_config_yaml_block_verbatim():
    """`hidden_size: null` (config.yaml:48) is resolved offline instead of raising: the reference fills it from the
    pretrained model's config (models.py:80-91 -> 384 for all-MiniLM-L6-v2 = 32 x the 12 heads), and the item table's
    width is checked against it."""
    import yaml

    import xfmr_rec_amd as X

    block = yaml.safe_load(REFERENCE_CONFIG_YAML_MODEL_BLOCK)
    conf = X.LightningConfig(**block)
    assert conf.hidden_size is None and conf.items_config["index_metric"] == "cosine"
    mod = X.RecommenderLightningModule(conf)
    mod.configure_model()
    c = mod.model.config
    assert (c.hidden_size, c.num_attention_heads, c.num_hidden_layers, c.intermediate_size, c.max_seq_length) == (384, 12, 1, 48, 32)
    names, shapes, _, total = X.models.flat_layout(384, 48, 32, 1)
    assert mod.model.flat.numel() == total
    # the MiniLM-wide table fits, another width is an error (models.py:336-345: no projection in between)
    mod.model.configure_embeddings({"embedding": torch.randn(10, 384), "item_id": [f"i{k}" for k in range(10)]})
    assert mod.model.embeddings.shape == (11, 384) and float(mod.model.embeddings[0].abs().sum()) == 0.0
    other = X.RecommenderModel(X.ModelConfig())
    with pytest.raises(ValueError, match="hidden_size"):
        other.configure_embeddings({"embedding": torch.randn(10, 128), "item_id": list(range(10))})
    # an unknown model name: hidden_size follows from the head count (head size 32)
    m2 = X.RecommenderModel(X.ModelConfig(pretrained_model_name="someone/unknown-model", num_attention_heads=4))
    assert m2.config.hidden_size == 128


def test_load_table_from_files(tmp_path):
    """any (V, H) table FILE -- .npy, .safetensors, .pt (weights only), items.parquet."""
    import numpy as np
    from safetensors.torch import save_file

    import xfmr_rec_amd as X

    V, H = 9, 64
    w = torch.randn(V, H)
    padded = torch.cat([torch.zeros(1, H), w])

    def fresh():
        return X.RecommenderModel(X.ModelConfig(hidden_size=H, num_attention_heads=2))

    np.save(tmp_path / "t.npy", w.numpy())
    m = fresh(); m.load_table(tmp_path / "t.npy")
    assert torch.equal(m.embeddings, padded)
    np.save(tmp_path / "p.npy", padded.numpy())  # already carries the padding row
    m = fresh(); m.load_table(tmp_path / "p.npy")
    assert torch.equal(m.embeddings, padded)
    save_file({"embedding": w}, str(tmp_path / "t.safetensors"))
    m = fresh(); m.load_table(tmp_path / "t.safetensors")
    assert torch.equal(m.embeddings, padded)
    torch.save({"embeddings.weight": padded}, tmp_path / "t.pt")  # the reference module's own key
    m = fresh(); m.load_table(tmp_path / "t.pt")
    assert torch.equal(m.embeddings, padded)
    import pyarrow as pa
    import pyarrow.parquet as pq

    ids = [f"movie{k}" for k in range(V)]
    pq.write_table(pa.table({"item_id": ids, "item_text": ["x"] * V, "embedding": [r.tolist() for r in w]}),
                   str(tmp_path / "items.parquet"))
    m = fresh(); m.load_table(tmp_path / "items.parquet")
    assert torch.equal(m.embeddings, padded) and int(m.id2idx["movie3"]) == 4
    with pytest.raises(ValueError, match="hidden_size"):
        np.save(tmp_path / "w.npy", torch.randn(V, 32).numpy()); fresh().load_table(tmp_path / "w.npy")
    with pytest.raises(ValueError, match="unsupported file type"):
        fresh().load_table(tmp_path / "t.csv")


def test_model_state_dict_roundtrip_on_cpu():
    import xfmr_rec_amd as X
    from oracle import encoder as enc

    cfg = X.ModelConfig(hidden_size=64, num_attention_heads=2, intermediate_size=128, num_hidden_layers=2, max_seq_length=20)
    m = X.RecommenderModel(cfg)
    ref = enc.init_params(64, 2, 128, 20, seed=5)
    assert set(ref) == set(m.encoder_state_dict())
    m.load_encoder_state_dict(ref)
    for k, v in m.encoder_state_dict().items():
        assert v.shape == ref[k].shape and torch.equal(v, ref[k]), k
    assert sum(p.numel() for p in m.parameters()) == sum(v.numel() for v in ref.values())
    # HF init statistics: weights ~ N(0, 0.02), LayerNorm = (1, 0), biases = 0
    fresh = X.RecommenderModel(cfg, seed=1).encoder_state_dict()
    assert abs(fresh["encoder.layer.0.intermediate.dense.weight"].std().item() - 0.02) < 2e-3
    assert torch.all(fresh["encoder.layer.1.output.LayerNorm.weight"] == 1)
    assert torch.all(fresh["encoder.layer.1.output.dense.bias"] == 0)


# ----------------------------------------------------------------------------------------- N > 1 (gloo, CPU)
def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _ddp_worker(rank, world, port, out_dir, nL=2):
    sys.path.insert(0, str(ROOT))
    sys.path.insert(0, str(ROOT / "transformer-recommenders_amd"))
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    import xfmr_rec_amd as X
    from helpers import ragged_batch, unit_table
    from oracle import model as OM
    from xfmr_rec_amd import distributed as D

    r, _local, w = D.init_process_group_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    H, A, I, L, V, B = 64, 2, 64, 12, 40, max(6, 2 * world)
    cfg = X.ModelConfig(hidden_size=H, num_attention_heads=A, intermediate_size=I, num_hidden_layers=nL, max_seq_length=L)
    model = X.RecommenderModel(cfg, seed=3)  # same seed on every rank == replicated weights
    table = unit_table(V, H)
    batch, _ = ragged_batch(B, L, V, seed=1)
    rows = list(D.shard_rows(B, rank, world))
    shard = {k: v[rows] for k, v in batch.items()}
    # the per-rank gradient comes from the CPU oracle here (the device path needs a GPU); the glue under test
    # is the flat layout + the single SUM all-reduce + the 1/world scale folded into AdamW
    params = {k: v.clone().requires_grad_(True) for k, v in model.encoder_state_dict().items()}
    out = OM.compute_losses(params, table, shard, num_heads=A, max_seq_length=L, loss_cfg={}, kinds=("InfoNCELoss",),
                            with_stats=False)
    out["loss/InfoNCELoss"].backward()
    flat_grad = torch.zeros_like(model.flat)
    for name, shape, off in zip(model._names, model._shapes, model._offsets):
        flat_grad[off : off + params[name].numel()] = params[name].grad.flatten()
    local = flat_grad.clone()
    D.allreduce_flat_grad_(flat_grad)
    # the same exchange in two halves (upper layers first: on a GPU that half runs underneath the lower layers' backward)
    halved = D.HalvedAllReduce(model)
    two = halved.reduce_(local.clone())
    torch.save({"local": local, "reduced": flat_grad, "rows": rows, "halved": two, "boundary": halved.boundary},
               os.path.join(out_dir, f"r{rank}.pt"))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,nL", [(2, 2), (8, 3), (2, 1)])
def test_data_parallel_gloo_world2(tmp_path, world, nL):
    """World sizes 2 and 8 (the node the scaling bench runs on); HalvedAllReduce with an even layer count, an odd one
    (boundary = layer nL // 2's first tensor) and a single layer (boundary = 0: the "tail" is the whole buffer and no head
    message is sent, distributed.py: `if b > 0`)."""
    port = _free_port()
    mp.start_processes(_ddp_worker, args=(world, port, str(tmp_path), nL), nprocs=world, join=True, start_method="spawn")
    r = [torch.load(tmp_path / f"r{i}.pt", weights_only=True) for i in range(world)]
    B = max(6, 2 * world)
    assert sorted(sum((x["rows"] for x in r), [])) == list(range(B))  # every row on exactly one rank
    total = sum(x["local"].double() for x in r).float()
    for i in range(world):  # (fp32 sums of 8 addends depend on the collective's order: a few ulp)
        torch.testing.assert_close(r[i]["reduced"], total, rtol=1e-5, atol=1e-6)
    assert not torch.allclose(r[0]["local"], r[1]["local"])  # shards really differ (negatives stay rank-local)
    # two halves == one message: bit for bit for two ranks (a + b is exact either way), to summation order for eight
    # (a ring all-reduce splits its message into per-rank chunks, so an element's order of addition depends on the length)
    for i in range(world):
        if world == 2:
            assert torch.equal(r[i]["halved"], r[i]["reduced"])
        else:
            torch.testing.assert_close(r[i]["halved"], r[i]["reduced"], rtol=1e-5, atol=1e-6)
        assert torch.equal(r[i]["halved"], r[0]["halved"])  # every replica holds the same bits
    import xfmr_rec_amd as X

    names, shapes, offsets, n_total = X.models.flat_layout(64, 64, 12, nL)
    if nL == 1:
        assert r[0]["boundary"] == 0
    else:
        first = f"encoder.layer.{nL // 2}.attention.self.query.weight"
        assert r[0]["boundary"] == offsets[names.index(first)] and 0 < r[0]["boundary"] < n_total


def test_saved_directory_is_a_loadable_hf_bert_with_the_sentence_transformer_layout(tmp_path):
    """RecommenderModel.save writes what SentenceTransformer.save writes for the reference's model (models.py:104-149,
    261-269): transformers' own BertModel.from_pretrained must load the directory strictly and compute the same
    function as the oracle encoder from the same tensors; the Pooling / Normalize modules round-trip."""
    import json

    import torch
    from transformers import BertModel

    from oracle import encoder as enc
    from xfmr_rec_amd.models import ModelConfig, read_sentence_transformer_dir, write_sentence_transformer_dir

    H, A, I, nL, L = 64, 2, 96, 2, 12
    params = enc.init_params(H, nL, I, L, seed=3)
    cfg = ModelConfig(vocab_size=1, hidden_size=H, num_hidden_layers=nL, num_attention_heads=A, intermediate_size=I,
                      max_seq_length=L, pooling_mode="max", is_normalized=True)
    write_sentence_transformer_dir(tmp_path, cfg, params)
    mods = json.loads((tmp_path / "modules.json").read_text())
    assert [m["type"].rsplit(".", 1)[1] for m in mods] == ["Transformer", "Pooling", "Normalize"]
    assert json.loads((tmp_path / "1_Pooling" / "config.json").read_text())["pooling_mode_max_tokens"] is True
    hf = BertModel.from_pretrained(str(tmp_path), local_files_only=True).eval()
    assert hf.config.is_decoder and hf.config.max_position_embeddings == L
    g = torch.Generator().manual_seed(0)
    x = torch.randn(2, L, H, generator=g)
    mask = torch.ones(2, L, dtype=torch.long)
    mask[1, 7:] = 0
    with torch.no_grad():
        want = hf(inputs_embeds=x, attention_mask=mask).last_hidden_state
        got = enc.encoder_forward(params, x, mask, A)
    torch.testing.assert_close(got[mask.bool()], want[mask.bool()], rtol=1e-5, atol=1e-5)
    cfg2, state = read_sentence_transformer_dir(tmp_path)
    assert cfg2.pooling_mode == "max" and cfg2.is_normalized and cfg2.max_seq_length == L
    for k, v in params.items():
        assert torch.equal(state[k], v.detach()), k


def test_bench_presets_name_the_baseline_configs(monkeypatch):
    """bench.py's --preset flags reproduce the shapes of BASELINE.json's configs 2-5 (the default line is config 2)."""
    import importlib
    import sys

    bench = importlib.import_module("bench")
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse()
    assert (a.items, a.seq_len, a.hidden, a.layers, a.loss, a.batch, a.negatives) == (3883, 200, 128, 4, "InfoNCELoss", 512, "in_batch")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--preset", "config3"])
    assert bench.parse().loss == "PairwiseLogisticLoss"
    monkeypatch.setattr(sys, "argv", ["bench.py", "--preset", "config4"])
    a = bench.parse()
    assert (a.items, a.hidden, a.layers, a.inter, a.negatives) == (27278, 256, 6, 1024, "catalogue")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--preset", "config5"])
    a = bench.parse()
    assert (a.items, a.seq_len, a.hidden, a.loss) == (1_000_000, 512, 256, "AlignmentContrastiveLoss")
    assert bench.METRIC.startswith("user-sequences/sec")


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus N` started as ONE process must start N ranks itself (torch.distributed.run, one process
    per GPU) and relay rank 0's line; a world size that disagrees with --gpus is an error, never a silent 1-GPU run
    reported as N. XFMR_BENCH_DRY=1 runs the rank control flow (rendezvous, barrier, SUM all-reduce, MAX timing) over
    gloo without a GPU."""
    import json
    import subprocess

    env = dict(os.environ, XFMR_BENCH_DRY="1", OMP_NUM_THREADS="1")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE line, printed by rank 0
    assert lines[0]["n_gpus"] == 2 and lines[0]["allreduce_ok"] is True and lines[0]["dry"] is True
    assert lines[0]["exchange"]["rccl_world"] == 2
    # the node the driver's scaling bench uses: 8 ranks
    r8 = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8"], env=env, capture_output=True, text=True,
                        timeout=600)
    assert r8.returncode == 0, r8.stderr[-2000:]
    l8 = [json.loads(l) for l in r8.stdout.splitlines() if l.startswith("{")]
    assert len(l8) == 1 and l8[0]["n_gpus"] == 8 and l8[0]["exchange"]["rccl_world"] == 8 and l8[0]["allreduce_ok"] is True
    assert l8[0]["scaling"] == "weak"
    # launched as a single rank of a 1-rank world but asked to report 2 GPUs: refused
    bad = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert bad.returncode != 0 and "WORLD_SIZE=1" in bad.stderr


# ----------------------------------------------------------------------------------------- drop-in surface (CPU)
def _small_module(seed=0):
    import xfmr_rec_amd as X

    conf = X.LightningConfig(hidden_size=64, num_attention_heads=2, intermediate_size=96, num_hidden_layers=2,
                             max_seq_length=12)
    mod = X.RecommenderLightningModule(conf)
    mod.model = X.RecommenderModel(conf, device="cpu", seed=seed)
    return mod


def test_lightning_state_dict_roundtrip_and_reference_keyed_checkpoint():
    """state_dict() -> load_state_dict() must reproduce the flat parameter buffer bit for bit (resume / load of a
    checkpoint), and a reference-keyed dict (trainer.py:352-362: every HF BERT tensor under model.model.0.auto_model.,
    including word_embeddings and the pooler, without the table) must load too."""
    from oracle import encoder as enc

    a, b = _small_module(seed=1), _small_module(seed=2)
    assert not torch.equal(a.model.flat, b.model.flat)
    sd = a.state_dict()
    assert "model.flat" not in sd and "model.embeddings.weight" not in sd
    assert all(k.startswith("model.model.0.auto_model.") for k in sd)
    res = b.load_state_dict(sd)  # strict
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(a.model.flat, b.model.flat)
    # a checkpoint written by the reference: HF BertModel keys incl. the tensors this build does not keep
    ref = {f"model.model.0.auto_model.{k}": v for k, v in enc.init_params(64, 2, 96, 12, seed=7).items()}
    ref["model.model.0.auto_model.embeddings.word_embeddings.weight"] = torch.zeros(1, 64)
    ref["model.model.0.auto_model.pooler.dense.weight"] = torch.zeros(64, 64)
    ref["model.model.0.auto_model.pooler.dense.bias"] = torch.zeros(64)
    c = _small_module(seed=3)
    res = c.load_state_dict(ref, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in c.model.encoder_state_dict().items():
        assert torch.equal(v, ref["model.model.0.auto_model." + k]), k
    # genuinely missing / unknown tensors are reported, and raise when strict
    part = dict(ref)
    gone = "model.model.0.auto_model.encoder.layer.1.output.dense.weight"
    del part[gone]
    part["model.model.0.auto_model.encoder.layer.9.output.dense.weight"] = torch.zeros(64, 96)
    with pytest.raises(RuntimeError, match="missing keys"):
        c.load_state_dict(part, strict=True)
    res = c.load_state_dict(part, strict=False)  # what Lightning does (strict_loading = False, trainer.py:129)
    assert res.missing_keys == [gone] and len(res.unexpected_keys) == 1
    bad = dict(ref)
    bad[gone] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError, match="size mismatch"):
        c.load_state_dict(bad)


def test_configure_embeddings_builds_the_padded_table_and_id2idx():
    """models.py:234-259: a zero padding row is inserted in front (item i of the dataset is row i + 1), the embedding
    width must equal hidden_size (no projection: models.py:336-345), id2idx maps item_id -> row; both a
    datasets.Dataset (the reference's type) and a plain mapping are accepted; a second call changes nothing."""
    import datasets
    import numpy as np

    import xfmr_rec_amd as X

    g = torch.Generator().manual_seed(0)
    emb = torch.randn(7, 64, generator=g)
    ids = [f"m{i}" for i in (5, 3, 11, 2, 8, 13, 1)]
    cfg = X.ModelConfig(hidden_size=64, num_attention_heads=2, intermediate_size=64, num_hidden_layers=1, max_seq_length=8)
    for ds in (datasets.Dataset.from_dict({"item_id": ids, "embedding": emb.tolist()}),
               {"item_id": ids, "embedding": emb.numpy()}):
        m = X.RecommenderModel(cfg, device="cpu")
        m.configure_embeddings(ds)
        assert m.embeddings.shape == (8, 64) and m.embeddings.dtype == torch.float32
        assert torch.all(m.embeddings[0] == 0)
        torch.testing.assert_close(m.embeddings[1:], emb, rtol=1e-6, atol=1e-7)
        assert [int(m.id2idx[i]) for i in ids] == list(range(1, 8))
        before = m.embeddings
        m.configure_embeddings({"item_id": ["x"], "embedding": np.zeros((1, 64), np.float32)})  # already configured
        assert m.embeddings is before and int(m.id2idx["m5"]) == 1
    m = X.RecommenderModel(cfg, device="cpu")
    with pytest.raises(ValueError, match="width 32 != hidden_size 64"):
        m.configure_embeddings({"item_id": ids, "embedding": np.zeros((7, 32), np.float32)})


def test_structured_candidates_behave_like_the_dense_tensor_they_stand_for():
    """compute_embeds returns the (Np, 1+N, H) candidates structured (models.py:408-416 would materialise O(N^2 H) bytes);
    a caller of the reference that looks at the tensor -- shape, size, dim, dtype, indexing rows / columns -- gets what
    the dense tensor would give."""
    from xfmr_rec_amd.losses import CatalogCandidates, SharedNegatives

    g = torch.Generator().manual_seed(0)
    table = torch.randn(20, 8, generator=g)
    pos, neg = torch.tensor([3, 7, 7, 1, 19]), torch.tensor([2, 5, 7, 11])
    c = SharedNegatives(table, torch.ones(20), pos, neg)
    dense = torch.cat([table[pos][:, None], table[neg][None].expand(5, -1, -1)], 1)
    assert c.shape == dense.shape and c.size(1) == 5 and c.dim() == 3 and len(c) == 5 and c.dtype == dense.dtype
    assert torch.equal(c.materialize(), dense)
    assert torch.equal(c[2], dense[2]) and torch.equal(c[-1], dense[-1]) and torch.equal(c[1:4], dense[1:4])
    assert torch.equal(c[:, 0], dense[:, 0]) and torch.equal(c[2, 1:], dense[2, 1:]) and torch.equal(c[[0, 4], 2], dense[[0, 4], 2])
    k = CatalogCandidates(table, torch.ones(20), 3)
    assert k.shape == (3, 20, 8) and torch.equal(k[1], table) and torch.equal(k[:, 4], table[4][None].expand(3, -1))


def test_row_tile_plan_of_the_whole_row_kernels(native):
    """gemm.hip xf_plan_row_tiles: the plain map by default ((M + 63) / 64 tiles); with XFMR_ROW_TILES_SHORT=1 (an experiment
    switch read once per process: measured, not kept -- DESIGN.md section 7) the rows left over after whole 64-row tiles
    per CU, when few (<= 32 per CU), become one short tile per CU: 102 400 rows on 256 CUs = 6 x 256 full tiles + 256 tiles
    of 16 rows instead of 1600 tiles with a seventh on 64 CUs."""
    import subprocess

    code = ("import ctypes as C, sys; lib = C.CDLL(sys.argv[1]); lib.xf_ln_row_tiles.restype = C.c_int; "
            "lib.xf_ln_row_tiles.argtypes = [C.c_int64]; "
            "print([lib.xf_ln_row_tiles(m) for m in (102400, 25600, 300, 64 * 256 * 3, 64 * 256 * 2 + 1)])")
    env = {k: v for k, v in os.environ.items() if k != "XFMR_ROW_TILES_SHORT"}
    plain = subprocess.run([sys.executable, "-c", code, str(native.LIB_PATH)], env=env, capture_output=True, text=True, check=True)
    assert eval(plain.stdout) == [1600, 400, 5, 768, 513]
    short = subprocess.run([sys.executable, "-c", code, str(native.LIB_PATH)], env=env | {"XFMR_ROW_TILES_SHORT": "1"},
                           capture_output=True, text=True, check=True)
    # 16 rows left per CU -> 256 short tiles; 36 per CU -> plain; < 1 tile per CU -> plain; nothing left -> plain; 1 row left
    assert eval(short.stdout) == [6 * 256 + 256, 400, 5, 768, 512 + 1]
