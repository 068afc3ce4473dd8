"""Drop-in shim for `from QAViT import QAViT, QAViTConfig` (the v1 block)."""
from qavit_amd import QAViTConfig  # noqa: F401
from qavit_amd import QAViT as _QAViT


class QAViT(_QAViT):
    def __init__(self, config):
        super().__init__(config, variant="v1")
