import importlib
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # GPU tests are selected explicitly with -m gpu; without a GPU they are skipped, never silently passed
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _seed_per_test(request):
    """Every test starts from the same generator state (a function of its own name), so a comparison that draws random upstream
    gradients does not depend on which tests ran before it."""
    import zlib
    seed = zlib.crc32(request.node.name.encode()) & 0x7FFFFFFF
    torch.manual_seed(seed)
    np.random.seed(seed & 0xFFFF)
    yield


class _Golden:
    """golden_v1.npz (make_golden.py) + golden_v2.npz (make_golden_v2.py) behind one mapping; tags do not collide."""

    def __init__(self, *paths):
        self._z = [np.load(p) for p in paths]
        self.files = [k for z in self._z for k in z.files]

    def __getitem__(self, key):
        for z in self._z:
            if key in z.files:
                return z[key]
        raise KeyError(key)


@pytest.fixture(scope="session")
def golden():
    gd = os.path.join(ROOT, "tests", "golden")
    return _Golden(os.path.join(gd, "golden_v1.npz"), os.path.join(gd, "golden_v2.npz"))


@pytest.fixture(scope="session")
def golden_r3():
    """bf16 envelope of the reference itself (tests/golden/make_golden_r3.py): per-tensor deviation of its autocast(bfloat16) step from
    its fp32 step."""
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_r3.npz"))


@pytest.fixture(scope="session")
def Q():
    import qavit_amd
    return qavit_amd


@pytest.fixture(scope="session")
def oracle():
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    return importlib.import_module("qavit_oracle")


MODELS = {
    # tag -> (builder(Q, **cfg_overrides), oracle forward name, oracle variant, label smoothing)
    "c100": (lambda Q, **kw: Q.HQAViT(Q.HQAViTConfig(**kw)), "hqavit_forward", "hqa", 0.12),
    "tin": (lambda Q, **kw: Q.HQAViT(Q.HQAViTTinyINConfig(**kw)), "hqavit_forward", "hqa", 0.12),
    "q32": (lambda Q, **kw: Q.QAViT(Q.qavit32_config(**kw), "v1"), "qavit_forward", "v1", 0.1),
    "v2_32": (lambda Q, **kw: Q.QAViT(Q.qavit32_config(**kw), "v2"), "qavit_forward", "v2", 0.1),
    # golden_v2.npz: the ConvNeXt-Tiny style stem (HQAViTv2_CIFAR100.py) and QA-ViT v1 / v2 at 224 px (N = 196)
    "c100v2": (lambda Q, **kw: Q.HQAViT(Q.HQAViTConfig(**kw), stem="v2"), "hqavit_forward", "hqa", 0.12),
    "q224": (lambda Q, **kw: Q.QAViT(Q.QAViTConfig(**kw), "v1"), "qavit_forward", "v1", 0.1),
    "v2_224": (lambda Q, **kw: Q.QAViT(Q.QAViTConfig(**kw), "v2"), "qavit_forward", "v2", 0.1),
}
HQA_TAGS = ("c100", "tin", "c100v2")


def sig(t: torch.Tensor) -> np.ndarray:
    """Same activation signature as tests/golden/make_golden.py."""
    t = t.detach().float().cpu()
    flat = t.reshape(t.shape[0], -1)
    corner = flat[:2, :96].reshape(-1)
    stride = max(1, flat.shape[1] // 64)
    strided = flat[:, ::stride][:, :64].reshape(-1)
    mom = torch.stack([t.mean(), t.std(), t.abs().max(), t.abs().mean()])
    return torch.cat([corner, strided, mom]).numpy().astype(np.float32)


def max_rel(a, b) -> float:
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def zero_by_construction(name: str) -> bool:
    """Parameters whose gradient is identically zero in exact arithmetic, so the reference's recorded value is pure
    round-off: a bias directly followed by a normalisation that removes it (LayerNorm over the biased axis'
    complement, train-mode BatchNorm) or by a softmax over the axis it is constant along."""
    return (name.endswith("token_upmix.upsample_attn.bias")          # LN over channels removes a per-token constant
            or name.endswith("token_learner.attention.1.bias")       # softmax over tokens ignores a per-column constant
            or (name.startswith("cnn_stem.") and name.endswith(".0.bias")))   # conv bias before train-mode BatchNorm


# ---------------------------------------------------------------------------------------------------
# Host replica of the device counter RNG (qa-vit_amd/csrc/common.cuh: mix32 / rng_key / rng_uniform and
# attn_shared.h: attn_drop_pkey / attn_drop_factor), so a test can compute the EXACT dropout mask a kernel used.
# ---------------------------------------------------------------------------------------------------
_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x):
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def rng_key(seed: int, step: int, site: int):
    inner = (np.uint64(step & 0xFFFFFFFF) * np.uint64(0x9E3779B9) + np.uint64(site) * np.uint64(0x85EBCA6B) + np.uint64(0x68E31DA4)) & _M32
    return _mix32(np.uint64(seed & 0xFFFFFFFF) ^ _mix32(inner))


def rng_uniform(key, idx):
    idx = np.asarray(idx, dtype=np.uint64)
    h = _mix32(((idx * np.uint64(0x9E3779B9)) & _M32) ^ np.asarray(key, dtype=np.uint64))
    return ((h >> np.uint64(8)).astype(np.float64) * (1.0 / 16777216.0)).astype(np.float32)


def drop_keep(key, idx, p):
    """bool: element kept by common.cuh drop_factor(key, idx, p): elements 2m, 2m+1 share the hash of m (low / high 16 bits)."""
    idx = np.asarray(idx, dtype=np.uint64)
    h = _mix32((((idx >> np.uint64(1)) * np.uint64(0x9E3779B9)) & _M32) ^ np.asarray(key, dtype=np.uint64))
    u16 = np.where((idx & np.uint64(1)) == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    thr = np.uint64(int(np.float32(p) * np.float32(65536.0)))
    return u16 >= thr


def attn_keep_mask(seed: int, step: int, site: int, G: int, H: int, Nq: int, NK: int, p: float) -> np.ndarray:
    """bool [G, H, Nq, NK]: True where the attention probability of (group g, head h, query i, key j) is KEPT."""
    key = rng_key(seed, step, site)
    pid = np.arange(G * H, dtype=np.uint64)
    pkey = _mix32((key + pid * np.uint64(0x9E3779B9)) & _M32)                       # [G*H]
    i = np.arange(Nq, dtype=np.uint64)[:, None]
    j = np.arange(NK, dtype=np.uint64)[None, :]
    idx = (i << np.uint64(16)) | j                                                    # [Nq, NK]
    return drop_keep(pkey[:, None, None], idx[None], p).reshape(G, H, Nq, NK)
