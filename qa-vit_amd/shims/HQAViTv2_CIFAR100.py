"""Drop-in shim for `from HQAViTv2_CIFAR100 import HQAViT, HQAViTConfig` (ConvNeXt-Tiny style CNN stem, layer-scaled blocks)."""
from qavit_amd import HQAViTConfig, ModelEMA, TrainingConfig  # noqa: F401
from qavit_amd import HQAViT as _HQAViT


class HQAViT(_HQAViT):
    def __init__(self, config):
        super().__init__(config, stem="v2")
