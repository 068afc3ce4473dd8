"""Key-name-seeded deterministic weight filler (golden-vector item 1).

There is no pretrained checkpoint anywhere in the reference tree, and RNG-replaying the reference's
initialisers across two different module trees is brittle.  Instead every parameter / buffer is filled
from a generator seeded by ``crc32(canonical key)``, so the reference model (in ``make_golden.py``), the
CPU oracle and the HIP model all hold bit-identical fp32 weights without a weight fixture.

Iteration is over ``named_parameters()`` + ``named_buffers()`` (canonical, de-duplicated names), never
over ``state_dict()``: the bank is aliased under every attention branch and must be filled once.
"""
import math
import zlib

import torch


def _gen(key: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(key.encode()))
    return g


def fill_tensor(key: str, t: torch.Tensor) -> torch.Tensor:
    """Return the filler value for ``key`` with ``t``'s shape/dtype (computed on CPU, fp32)."""
    g = _gen(key)
    shape = tuple(t.shape)
    leaf = key.rsplit(".", 1)[-1]
    if not t.dtype.is_floating_point:                      # update_count, num_batches_tracked
        return torch.zeros(shape, dtype=t.dtype)
    r = torch.randn(shape, generator=g, dtype=torch.float32)
    if leaf == "running_var":
        v = r.abs() + 0.5
    elif leaf == "running_mean":
        v = 0.05 * r
    elif leaf in ("gamma", "scale", "beta"):               # ccf_ffn.gamma, dwconv.scale, rrcv.beta (init 0.1)
        v = 0.1 + 0.02 * r
    elif leaf == "fusion_weights":
        v = 1.0 + 0.1 * r
    elif leaf == "weight" and t.ndim == 1:                 # LayerNorm / BatchNorm weight
        v = 1.0 + 0.02 * r
    elif leaf == "weight" and t.ndim == 4:                 # Conv2d: kaiming fan_out scale (reference _init_weights)
        fan_out = shape[0] * shape[2] * shape[3] / 1.0
        # grouped conv: fan_out counts out_channels/groups... keep the reference's definition
        v = math.sqrt(2.0 / fan_out) * r
    else:                                                  # Linear weights, biases, embeddings, bank, E_k/E_v
        v = 0.02 * r
    return v.to(t.dtype)


@torch.no_grad()
def fill_module(model: torch.nn.Module) -> torch.nn.Module:
    """Fill every parameter and buffer of ``model`` in place."""
    for name, p in model.named_parameters():
        p.copy_(fill_tensor(name, p).to(p.device))
    for name, b in model.named_buffers():
        b.copy_(fill_tensor(name, b).to(b.device))
    return model
