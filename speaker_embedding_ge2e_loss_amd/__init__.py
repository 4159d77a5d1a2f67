"""MI355X-native GE2E loss: drop-in for gkv856/speaker_embedding_GE2E_loss's GE2ELoss."""
from .loss import GE2ELoss, HParams  # noqa: F401
from . import functional  # noqa: F401

__all__ = ["GE2ELoss", "HParams", "functional"]
