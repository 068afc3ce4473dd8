"""Drop-in shim for `from QAViTv2 import QAViT, QAViTConfig` (224 px / patch 16 by default; the v2 block)."""
from qavit_amd import QAViTConfig  # noqa: F401
from qavit_amd import QAViT as _QAViT


class QAViT(_QAViT):
    def __init__(self, config):
        super().__init__(config, variant="v2")
