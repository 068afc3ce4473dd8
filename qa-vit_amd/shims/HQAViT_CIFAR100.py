"""Drop-in shim: put this directory ahead of the reference on sys.path and `from HQAViT_CIFAR100 import HQAViT,
HQAViTConfig, ModelEMA, GradientMonitor, TrainingConfig` resolves to the MI355X implementation."""
from qavit_amd import HQAViT, HQAViTConfig, ModelEMA, TrainingConfig  # noqa: F401
from qavit_amd.harness import GradientMonitor  # noqa: F401
