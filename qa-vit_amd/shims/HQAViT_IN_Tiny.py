"""Drop-in shim for `from HQAViT_IN_Tiny import HQAViT, HQAViTConfig` (64x64, depth 12, 64 learned tokens)."""
from qavit_amd import HQAViT, ModelEMA, TrainingConfig  # noqa: F401
from qavit_amd import HQAViTTinyINConfig as HQAViTConfig  # noqa: F401
