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
