"""Drop-in shim for `from QAViTV2_EXTREME import QAViT, QAViTConfig, TrainingConfig` (test.py:16): the 32x32 v2 block
without depthwise bias (= the 'hqa' block variant)."""
from qavit_amd import QAViTConfig, TrainingConfig  # noqa: F401
from qavit_amd import QAViT as _QAViT


class QAViT(_QAViT):
    def __init__(self, config):
        super().__init__(config, variant="hqa")
