"""Config dataclasses with the reference's field names and defaults.

``HQAViTConfig``       = HQAViT_CIFAR100.py:42-78
``HQAViTTinyINConfig`` = HQAViT_IN_Tiny.py:48-84 (64x64, depth 12, 64 learned tokens, 200 classes)
``QAViTConfig``        = QAViT.py:36-56 (224/16 defaults)
``qavit32_config()``   = the 32x32 settings QAViTv2_CIFAR100.py:43-60 uses (BASELINE config 2)
"""
from dataclasses import dataclass
from typing import Tuple


@dataclass
class HQAViTConfig:
    img_size: int = 32
    patch_size: int = 4
    in_channels: int = 3
    num_classes: int = 100
    embed_dim: int = 192
    depth: int = 8
    num_heads: int = 4
    compress_ratio: int = 4
    bottleneck_ratio: int = 2
    mlp_ratio: float = 0.5
    global_bank_size: int = 16
    dropout: float = 0.1
    drop_path: float = 0.1
    window_size: int = 4
    dilation_factors: Tuple[int, ...] = (1, 2)
    landmark_pooling_stride: int = 2
    num_channel_groups: int = 6
    linformer_k: int = 32
    cnn_c2: int = 64
    cnn_c3: int = 128
    cnn_c4: int = 256
    rrcv_channels: int = 64
    rrcv_num_blocks: int = 1
    use_token_learner: bool = True
    num_learned_tokens: int = 16
    fusion_stages: Tuple[int, ...] = (2, 3, 4)


@dataclass
class HQAViTTinyINConfig(HQAViTConfig):
    img_size: int = 64
    num_classes: int = 200
    depth: int = 12
    drop_path: float = 0.2
    num_learned_tokens: int = 64


@dataclass
class QAViTConfig:
    img_size: int = 224
    patch_size: int = 16
    in_channels: int = 3
    num_classes: int = 100
    embed_dim: int = 192
    depth: int = 8
    num_heads: int = 4
    compress_ratio: int = 4
    bottleneck_ratio: int = 2
    mlp_ratio: float = 0.5
    global_bank_size: int = 16
    dropout: float = 0.1
    drop_path: float = 0.1
    window_size: int = 7
    dilation_factors: Tuple[int, ...] = (1, 2, 3)
    landmark_pooling_stride: int = 2
    num_channel_groups: int = 6
    linformer_k: int = 64
    # not a reference field: QAViT has no TokenLearner; kept so shared block code can ask
    use_token_learner: bool = False


def qavit32_config(**kw) -> QAViTConfig:
    """QA-ViT at 32x32 / patch 4 (N=64): SURVEY.md section 3.3."""
    base = dict(img_size=32, patch_size=4, window_size=4, dilation_factors=(1, 2), linformer_k=32)
    base.update(kw)
    return QAViTConfig(**base)
