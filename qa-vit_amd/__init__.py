"""qa-vit_amd: MI355X-native (gfx950) implementation of the QA-ViT / HQA-ViT hot path.

The directory name is the one the build contract fixes; it is not a valid Python identifier, so import it
as ``qavit_amd`` (the repo-root shim ``qavit_amd.py`` aliases it) or via
``importlib.import_module("qa-vit_amd")``.

Public surface (mirrors the reference's importable names):
    HQAViT, HQAViTConfig, HQAViTTinyINConfig, QAViT, QAViTConfig, qavit32_config,
    ModelEMA, Trainer, DataParallel, fill_module
"""
from .config import HQAViTConfig, HQAViTTinyINConfig, QAViTConfig, qavit32_config  # noqa: F401
from .filler import fill_module, fill_tensor  # noqa: F401
from .models import HQAViT, QAViT  # noqa: F401
from .harness import BatchStager, FineTuneConfig, GradientMonitor, ModelEMA, Trainer, TrainingConfig, mix_plan  # noqa: F401
from . import harness  # noqa: F401
from .parallel import DataParallel  # noqa: F401
from . import lib  # noqa: F401

__all__ = ["HQAViT", "HQAViTConfig", "HQAViTTinyINConfig", "QAViT", "QAViTConfig", "qavit32_config",
           "BatchStager", "FineTuneConfig", "GradientMonitor", "ModelEMA", "Trainer", "TrainingConfig", "mix_plan", "DataParallel", "fill_module", "fill_tensor", "lib"]
