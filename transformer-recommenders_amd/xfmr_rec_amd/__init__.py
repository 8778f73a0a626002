"""xfmr_rec_amd -- MI355X-native training path of yxtay/transformer-recommenders.

Host-side mirror of the reference's ``xfmr_rec.models`` / ``xfmr_rec.losses`` / ``xfmr_rec.trainer``
interfaces over hand-written gfx950 HIP kernels (``libxfmr_hip.so``, C ABI in ``include/xfmr_hip.h``).
"""

from .losses import LOSS_CLASSES, EmbedLoss, LossConfig, LossType  # noqa: F401
from .models import ModelConfig, RecommenderModel  # noqa: F401
from .trainer import FusedAdamW, GraphedStep, LightningConfig, RecommenderLightningModule, Trainer  # noqa: F401

__all__ = [
    "LOSS_CLASSES", "EmbedLoss", "LossConfig", "LossType", "ModelConfig", "RecommenderModel",
    "FusedAdamW", "GraphedStep", "LightningConfig", "RecommenderLightningModule", "Trainer",
]
