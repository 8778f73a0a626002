"""Constants of the training path (values of the reference's ``xfmr_rec/params.py:1-19``)."""

# model
PRETRAINED_MODEL_NAME = "sentence-transformers/all-MiniLM-L6-v2"
METRIC = {"name": "val/retrieval_normalized_dcg", "mode": "max"}
TOP_K = 20

# artefact layout (xfmr_rec/params.py:15-19)
ITEMS_TABLE_NAME = "items"
LANCE_DB_PATH = "lance_db"
TRANSFORMER_PATH = "transformer"
USERS_TABLE_NAME = "users"

# BERT defaults the reference never overrides (TF:models/bert/configuration_bert.py:44-63)
HIDDEN_DROPOUT_PROB = 0.1
ATTENTION_PROBS_DROPOUT_PROB = 0.1
LAYER_NORM_EPS = 1e-12
INITIALIZER_RANGE = 0.02

# Published configurations of the pretrained models a ModelConfig may name: what the reference's
# AutoModel.from_pretrained(name).config / AutoTokenizer.from_pretrained(name) would return for the fields it reads
# (xfmr_rec/models.py:69-91). This build is offline, so the values are recorded here (model cards on the HF hub).
KNOWN_PRETRAINED = {
    "sentence-transformers/all-MiniLM-L6-v2": dict(
        vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12, intermediate_size=1536,
        max_seq_length=512),
    "sentence-transformers/all-MiniLM-L12-v2": dict(
        vocab_size=30522, hidden_size=384, num_hidden_layers=12, num_attention_heads=12, intermediate_size=1536,
        max_seq_length=512),
    "sentence-transformers/all-mpnet-base-v2": dict(
        vocab_size=30527, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
        max_seq_length=512),
}
