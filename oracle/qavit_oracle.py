"""CPU oracle for the QA-ViT / HQA-ViT forward hot path.

*** TEST INFRASTRUCTURE -- NOT PRODUCT CODE. ***
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module, and only as the checker / the timed CPU baseline.  The product path (``qa-vit_amd``) never
imports it and fails loudly when the HIP extension is missing.

What it is: a functional (state_dict-keyed) fp32 PyTorch restatement of the reference algorithm, written
from the reference's behaviour, not copied from it.  Every function cites the reference file:line it
follows.  Arithmetic is floating point, so the primitives (``layer_norm``, ``gelu`` (erf), ``softmax``,
``scaled_dot_product_attention``, ``conv2d``) are PyTorch's own CPU kernels -- exactly what the reference
calls ("third-party arithmetic").

Pinning: ``tests/golden/make_golden.py`` imports the real reference (``/root/reference``; only possible
in the build container), injects the key-name-seeded weights of ``qa-vit_amd/filler.py`` and records
logits / loss / intermediates / bank state / gradient norms into ``tests/golden/*.npz``.
``tests/test_oracle_golden.py`` checks this oracle against those vectors (<=1e-5), so parity is PINNED.

Variants (``Variant``):
  * ``hqa``   -- HQAViT_CIFAR100.py / HQAViT_IN_Tiny.py / QAViTv2_CIFAR100.py block
  * ``v2``    -- QAViTv2.py block (= hqa + depthwise-conv bias)
  * ``v1``    -- QAViT.py block (no CCF-FFN norms/gamma/scale, bank clamp 0.1/1.0, fixed rate 0.01)
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


@dataclass
class Variant:
    name: str = "hqa"
    ccf_norm: bool = True      # CCFFFN has dwconv_norm / post_dwconv_norm / gamma (HQAViT_CIFAR100.py:678-712)
    dw_bias: bool = False      # depthwise conv bias (QAViT.py:556, QAViTv2.py:861)
    dw_scale: bool = True      # DepthwiseConv2d.scale (HQAViT_CIFAR100.py:668)
    bank_v1: bool = False      # QAViT.py:217-224 bank update rule


VARIANTS = {
    "hqa": Variant("hqa", True, False, True, False),
    "v2": Variant("v2", True, True, True, False),
    "v1": Variant("v1", False, True, False, True),
}


# --------------------------------------------------------------------------------------------------
# primitives
# --------------------------------------------------------------------------------------------------
def _ln(P: Dict[str, Tensor], pre: str, x: Tensor, eps: float = 1e-5) -> Tensor:
    return F.layer_norm(x, (x.shape[-1],), P[pre + ".weight"], P[pre + ".bias"], eps)


def _lin(P: Dict[str, Tensor], pre: str, x: Tensor) -> Tensor:
    return F.linear(x, P[pre + ".weight"], P.get(pre + ".bias"))


# Optional mask injection (tests only): ``set_masks(fn)`` installs ``fn(name, kind, shape, p) -> bool keep tensor or None``; every
# stochastic site below then multiplies by keep / (1 - p) -- the documented math of nn.Dropout, drop_path (:256-263) and SDPA's
# ``dropout_p`` -- with the caller's mask instead of torch's RNG.  ``name`` = the module path of the site's owner + a suffix
# (".proj" / ".attn" / ".dp1" ...), ``kind`` in {"drop", "path", "attn"}.  With no provider installed nothing changes.
_MASKS = None


def set_masks(fn) -> None:
    global _MASKS
    _MASKS = fn


def _given(name: Optional[str], kind: str, shape, p: float):
    if _MASKS is None or name is None:
        return None
    m = _MASKS(name, kind, tuple(shape), p)
    return None if m is None else torch.as_tensor(m)


def _drop(x: Tensor, p: float, train: bool, name: Optional[str] = None) -> Tensor:
    if not (train and p > 0.0):
        return x
    keep = _given(name, "drop", x.shape, p)
    if keep is not None:
        return x * keep.reshape(x.shape).to(x.dtype) / (1.0 - p)
    return F.dropout(x, p, train)


def _drop_path(x: Tensor, p: float, train: bool, name: Optional[str] = None) -> Tensor:
    """HQAViT_CIFAR100.py:256-263 -- per-sample Bernoulli keep, scaled by 1/keep."""
    if p == 0.0 or not train:
        return x
    keep = 1.0 - p
    given = _given(name, "path", (x.shape[0],), p)
    if given is not None:
        mask = given.reshape((x.shape[0],) + (1,) * (x.ndim - 1)).to(x.dtype)
    else:
        mask = (keep + torch.rand((x.shape[0],) + (1,) * (x.ndim - 1), dtype=x.dtype)).floor_()
    return x.div(keep) * mask


def _sdpa(q: Tensor, k: Tensor, v: Tensor, p: float, train: bool, name: Optional[str] = None) -> Tensor:
    """efficient_attention, HQAViT_CIFAR100.py:355-397: NaN in -> zeros, SDPA, NaN out -> zeros."""
    if torch.isnan(q).any() or torch.isnan(k).any() or torch.isnan(v).any():
        return torch.zeros_like(q)
    keep = _given(name, "attn", (q.shape[0], q.shape[1], q.shape[2], k.shape[2]), p) if (train and p > 0.0) else None
    if keep is not None:            # softmax(QK^T / sqrt(D)) * keep / (1 - p) @ V: scaled_dot_product_attention's dropout_p branch
        pr = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), -1)
        o = (pr * keep.to(pr.dtype) / (1.0 - p)) @ v
    else:
        o = F.scaled_dot_product_attention(q, k, v, dropout_p=p if train else 0.0)
    if torch.isnan(o).any():
        return torch.zeros_like(o)
    return o


def _grid(n: int) -> int:
    return int(math.sqrt(n))


# --------------------------------------------------------------------------------------------------
# global token bank (HQAViT_CIFAR100.py:275-321, QAViT.py:183-224)
# --------------------------------------------------------------------------------------------------
def bank_write(P, tokens: Tensor, var: Variant, train: bool, sync=None) -> None:
    """In-place, no-grad update of global_bank.global_k / global_v from a [B,N,C] token tensor.

    ``sync`` (optional) is called on the batch-mean update [2,S,C] before clamping: the data-parallel
    exact mode all-reduces (mean) there (SURVEY.md section 8e exception 1)."""
    if not train:
        return
    with torch.no_grad():
        tn = _ln(P, "global_bank.write_norm", tokens)
        comp = _lin(P, "global_bank.write_compression", tn)
        w = F.softmax(_lin(P, "global_bank.write_gate", tn), dim=1)            # softmax over tokens
        upd_k = torch.bmm(w.transpose(1, 2), comp).mean(0, keepdim=True)
        upd_v = torch.bmm(w.transpose(1, 2), tn).mean(0, keepdim=True)
        if sync is not None:
            both = sync(torch.cat([upd_k, upd_v], 0))
            upd_k, upd_v = both[0:1], both[1:2]
        gk, gv = P["global_bank.global_k"], P["global_bank.global_v"]
        if var.bank_v1:                                                        # QAViT.py:217-224
            gk.data.add_(0.01 * upd_k.clamp(-0.1, 0.1)).clamp_(-1.0, 1.0)
            gv.data.add_(0.01 * upd_v.clamp(-0.1, 0.1)).clamp_(-1.0, 1.0)
        else:                                                                  # HQAViT_CIFAR100.py:309-321
            cnt = P["global_bank.update_count"]
            rate = 0.005 if int(cnt) < 1000 else 0.01
            gk.data.add_(rate * upd_k.clamp(-0.05, 0.05)).clamp_(-0.5, 0.5)
            gv.data.add_(rate * upd_v.clamp(-0.05, 0.05)).clamp_(-0.5, 0.5)
            cnt += 1


def _bank_heads(P, nb: int, heads: int):
    gk, gv = P["global_bank.global_k"], P["global_bank.global_v"]             # [1,S,C]
    S, C = gk.shape[1], gk.shape[2]
    k = gk.expand(nb, -1, -1).reshape(nb, S, heads, C // heads).transpose(1, 2)
    v = gv.expand(nb, -1, -1).reshape(nb, S, heads, C // heads).transpose(1, 2)
    return k, v


def _linformer(P, pre: str, k: Tensor, v: Tensor):
    """LinformerCompression.forward, HQAViT_CIFAR100.py:332-352: pad/truncate to seq_len, E^T @ K."""
    Ek, Ev = P[pre + ".E_k"], P[pre + ".E_v"]                                  # [L, kc]
    L = Ek.shape[0]
    B, H, N, D = k.shape
    if N < L:
        k = F.pad(k, (0, 0, 0, L - N))
        v = F.pad(v, (0, 0, 0, L - N))
    elif N > L:
        k, v = k[:, :, :L], v[:, :, :L]
    kc = torch.matmul(Ek.T, k.reshape(B * H, L, D)).reshape(B, H, -1, D)
    vc = torch.matmul(Ev.T, v.reshape(B * H, L, D)).reshape(B, H, -1, D)
    return kc, vc


# --------------------------------------------------------------------------------------------------
# attention branches
# --------------------------------------------------------------------------------------------------
def swa(P, pre: str, x: Tensor, cfg, var: Variant, train: bool, sync=None) -> Tensor:
    """EfficientSpatialWindowAttention.forward, HQAViT_CIFAR100.py:441-469."""
    B, N, C = x.shape
    H = W = _grid(N)
    ws, heads = cfg.window_size, cfg.num_heads
    D = C // heads
    g = x.view(B, H, W, C)
    ph, pw = (ws - H % ws) % ws, (ws - W % ws) % ws                            # :424-428
    if ph or pw:
        g = F.pad(g, (0, 0, 0, pw, 0, ph))
    Hp, Wp = H + ph, W + pw
    nh, nw = Hp // ws, Wp // ws
    win = g.view(B, nh, ws, nw, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, C)
    BW, NW = win.shape[0], win.shape[1]
    qkv = _lin(P, pre + ".qkv", win).reshape(BW, NW, 3, heads, D).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    kc, vc = _linformer(P, pre + ".linformer", k, v)
    bk, bv = _bank_heads(P, BW, heads)
    o = _sdpa(q, torch.cat([kc, bk], 2), torch.cat([vc, bv], 2), cfg.dropout, train, pre + ".attn")
    o = o.transpose(1, 2).reshape(BW, NW, C)
    o = _drop(_lin(P, pre + ".proj", o), cfg.dropout, train, pre + ".proj")
    # window_reverse is called with the UNPADDED H, W (:466); identical when no padding happened
    o = o.view(B, H // ws, W // ws, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B, H * W, C)
    bank_write(P, _ln(P, pre + ".norm", o), var, train, sync)
    return o


def msda(P, pre: str, x: Tensor, cfg, var: Variant, train: bool, sync=None) -> Tensor:
    """EfficientMultiScaleDilatedAttention.forward, HQAViT_CIFAR100.py:496-532."""
    B, N, C = x.shape
    H = W = _grid(N)
    heads = cfg.num_heads
    D = C // heads
    g = x.view(B, H, W, C)
    multi = torch.cat([g[:, ::d, ::d, :].reshape(B, -1, C) for d in cfg.dilation_factors], 1)
    s = cfg.landmark_pooling_stride
    pooled = F.avg_pool1d(multi.transpose(1, 2), s, s).transpose(1, 2)         # :501
    NM = pooled.shape[1]
    kv = _lin(P, pre + ".qkv", pooled).reshape(B, NM, 3, heads, D).permute(2, 0, 3, 1, 4)
    kc, vc = _linformer(P, pre + ".linformer", kv[1], kv[2])                   # pad/trunc to 128 inside
    bk, bv = _bank_heads(P, B, heads)
    q = _lin(P, pre + ".qkv", x).reshape(B, N, 3, heads, D)[:, :, 0].permute(0, 2, 1, 3)   # :523
    o = _sdpa(q, torch.cat([kc, bk], 2), torch.cat([vc, bv], 2), cfg.dropout, train, pre + ".attn")
    o = o.transpose(1, 2).reshape(B, N, C)
    o = _drop(_lin(P, pre + ".proj", o), cfg.dropout, train, pre + ".proj")
    bank_write(P, _ln(P, pre + ".norm", o), var, train, sync)
    return o


def cga(P, pre: str, x: Tensor, cfg, var: Variant, train: bool, sync=None) -> Tensor:
    """EfficientChannelGroupAttention.forward, HQAViT_CIFAR100.py:559-595."""
    B, N, C = x.shape
    G, heads = cfg.num_channel_groups, cfg.num_heads
    cpg = C // G
    cc = C // 2
    ccg = cc // G
    dh = ccg // heads
    xf = x.view(B, N, G, cpg).permute(0, 2, 1, 3).reshape(B * G, N, cpg)
    q = _lin(P, pre + ".q_proj", xf).reshape(B * G, N, heads, dh).transpose(1, 2)
    k = _lin(P, pre + ".k_proj", xf).reshape(B * G, N, heads, dh).transpose(1, 2)
    v = _lin(P, pre + ".v_proj", xf).reshape(B * G, N, heads, dh).transpose(1, 2)
    gk, gv = P["global_bank.global_k"], P["global_bank.global_v"]
    S = gk.shape[1]
    # The reference projects the EXPANDED bank (:576-577).  That matters for autograd: linear's flatten of a
    # stride-0 view copies, so the saved input holds the forward-time bank, not the later in-place writes.
    bk = _lin(P, pre + ".bank_k_proj", gk.expand(B, -1, -1)).unsqueeze(1).expand(-1, G, -1, -1).reshape(B * G, S, heads, dh).transpose(1, 2)
    bv = _lin(P, pre + ".bank_v_proj", gv.expand(B, -1, -1)).unsqueeze(1).expand(-1, G, -1, -1).reshape(B * G, S, heads, dh).transpose(1, 2)
    o = _sdpa(q, torch.cat([k, bk], 2), torch.cat([v, bv], 2), cfg.dropout, train, pre + ".attn")
    o = o.transpose(1, 2).reshape(B, G, N, ccg).permute(0, 2, 1, 3).reshape(B, N, cc)
    o = _drop(_lin(P, pre + ".proj", o), cfg.dropout, train, pre + ".proj")
    bank_write(P, _ln(P, pre + ".norm", o), var, train, sync)
    return o


def cross(P, pre: str, x: Tensor, cfg, train: bool) -> Tensor:
    """CrossAttentionBranch.forward, HQAViT_CIFAR100.py:613-626 (no bank write)."""
    B, N, C = x.shape
    heads = cfg.num_heads
    D = C // heads
    q = _lin(P, pre + ".q_proj", x).reshape(B, N, heads, D).transpose(1, 2)
    gk, gv = P["global_bank.global_k"], P["global_bank.global_v"]
    S = gk.shape[1]
    # projected AFTER expansion, as the reference does (:617-619) -- see the note in cga()
    k = _lin(P, pre + ".k_proj", gk.expand(B, -1, -1)).reshape(B, S, heads, D).transpose(1, 2)
    v = _lin(P, pre + ".v_proj", gv.expand(B, -1, -1)).reshape(B, S, heads, D).transpose(1, 2)
    o = _sdpa(q, k, v, cfg.dropout, train, pre + ".attn")
    o = o.transpose(1, 2).reshape(B, N, C)
    return _drop(_lin(P, pre + ".proj", o), cfg.dropout, train, pre + ".proj")


# --------------------------------------------------------------------------------------------------
# fusion / FFN / block
# --------------------------------------------------------------------------------------------------
def ccf_ffn(P, pre: str, x: Tensor, cfg, var: Variant, train: bool) -> Tensor:
    """CCFFFN.forward, HQAViT_CIFAR100.py:700-712 (v1: QAViT.py:575-582)."""
    B, N, C = x.shape
    H = W = _grid(N)
    h = F.gelu(_lin(P, pre + ".fc1", x))
    if var.ccf_norm:
        h = _ln(P, pre + ".dwconv_norm", h)
    hc = h.shape[-1]
    img = h.transpose(1, 2).reshape(B, hc, H, W)                               # DepthwiseConv2d :670-675
    img = F.conv2d(img, P[pre + ".dwconv.dwconv.weight"], P.get(pre + ".dwconv.dwconv.bias"),
                   padding=1, groups=hc)
    if var.dw_scale:
        img = img * P[pre + ".dwconv.scale"]
    h = img.flatten(2).transpose(1, 2)
    if var.ccf_norm:
        h = _ln(P, pre + ".post_dwconv_norm", h)
    h = _drop(_lin(P, pre + ".fc2", h), cfg.dropout, train, pre + ".fc2")
    if var.ccf_norm:
        h = h * P[pre + ".gamma"]
    return h


def quad_block(P, pre: str, x: Tensor, cfg, var: Variant, train: bool, dp: float, taps=None, sync=None) -> Tensor:
    """QuadAttentionBlock.forward, HQAViT_CIFAR100.py:1071-1085.  Branch order matters in train mode:
    swa, msda and cga each mutate the bank the next branch reads."""
    xn = _ln(P, pre + ".norm1", x)
    b_swa = swa(P, pre + ".swa", xn, cfg, var, train, sync)
    b_msda = msda(P, pre + ".msda", xn, cfg, var, train, sync)
    b_cga = cga(P, pre + ".cga", xn, cfg, var, train, sync)
    b_cross = cross(P, pre + ".cross_attn", xn, cfg, train)
    if taps is not None:
        taps[pre + ".swa"], taps[pre + ".msda"] = b_swa, b_msda
        taps[pre + ".cga"], taps[pre + ".cross_attn"] = b_cga, b_cross
    outs = []
    for name, t in (("swa", b_swa), ("msda", b_msda), ("cga", b_cga), ("cross", b_cross)):
        outs.append(_lin(P, f"{pre}.compress_{name}", _ln(P, f"{pre}.norm_{name}", t)))
    fw = F.softmax(P[pre + ".fusion.fusion_weights"], dim=0)                   # HybridFusion :637-640
    fused = torch.cat([o * fw[i] for i, o in enumerate(outs)], -1)
    h = _drop(F.gelu(_lin(P, pre + ".bottleneck_mlp.fc1", fused)), cfg.dropout, train, pre + ".bottleneck_mlp.fc1")   # :651-656
    h = _drop(_lin(P, pre + ".bottleneck_mlp.fc2", h), cfg.dropout, train, pre + ".bottleneck_mlp.fc2")
    x = x + _drop_path(h, dp, train, pre + ".dp1")
    x = x + _drop_path(ccf_ffn(P, pre + ".ccf_ffn", _ln(P, pre + ".norm2", x), cfg, var, train), dp, train, pre + ".dp2")
    return x


def token_learner(P, pre: str, x: Tensor) -> Tensor:
    """TokenLearner.forward, HQAViT_CIFAR100.py:985-1002: softmax over the N axis."""
    s = _lin(P, pre + ".attention.1", _ln(P, pre + ".attention.0", x))         # [B,N,M]
    s = F.softmax(s, dim=1)
    return torch.bmm(s.transpose(1, 2), x)


def token_upmix(P, pre: str, xc: Tensor) -> Tensor:
    """TokenUpMix.forward, HQAViT_CIFAR100.py:1016-1031: Linear(M->N) on the token axis, then LN."""
    up = _lin(P, pre + ".upsample_attn", xc.transpose(1, 2)).transpose(1, 2)
    return _ln(P, pre + ".norm", up)


def tl_block(P, pre: str, x: Tensor, cfg, var: Variant, train: bool, dp: float, taps=None, sync=None) -> Tensor:
    """QuadBlockWithTokenLearner.forward, HQAViT_CIFAR100.py:1104-1123 (no skip connection)."""
    if not cfg.use_token_learner:
        return quad_block(P, pre + ".quad_block", x, cfg, var, train, dp, taps, sync)
    xc = token_learner(P, pre + ".token_learner", x)
    xc = quad_block(P, pre + ".quad_block", xc, cfg, var, train, dp, taps, sync)
    return token_upmix(P, pre + ".token_upmix", xc)


def patch_embed(P, x: Tensor, cfg) -> Tensor:
    """PatchEmbed.forward, HQAViT_CIFAR100.py:1136-1138."""
    t = F.conv2d(x, P["patch_embed.proj.weight"], P["patch_embed.proj.bias"], stride=cfg.patch_size)
    return _ln(P, "patch_embed.norm", t.flatten(2).transpose(1, 2))


# --------------------------------------------------------------------------------------------------
# CNN lateral path (second tier)
# --------------------------------------------------------------------------------------------------
def _bn(P, pre: str, x: Tensor, train: bool) -> Tensor:
    return F.batch_norm(x, P[pre + ".running_mean"], P[pre + ".running_var"], P[pre + ".weight"],
                        P[pre + ".bias"], train, 0.1, 1e-5)


def _convnext(P, pre: str, x: Tensor, dp: float = 0.0, train: bool = False) -> Tensor:
    """ConvNeXtBlock.forward, HQAViT_CIFAR100.py:729-739 (drop_path is Identity: default 0) and its layer-scaled
    form HQAViTv2_CIFAR100.py:736-750 (``gamma`` present in the state dict; drop path ``dp``)."""
    h = F.conv2d(x, P[pre + ".dwconv.weight"], P[pre + ".dwconv.bias"], padding=3, groups=x.shape[1])
    h = h.permute(0, 2, 3, 1)
    h = _ln(P, pre + ".norm", h, 1e-6)
    h = _lin(P, pre + ".pwconv2", F.gelu(_lin(P, pre + ".pwconv1", h)))
    if pre + ".gamma" in P:
        h = P[pre + ".gamma"] * h
    return x + _drop_path(h.permute(0, 3, 1, 2), dp, train)


def cnn_stem(P, x: Tensor, train: bool):
    """CNNStemModel.forward, HQAViT_CIFAR100.py:779-793."""
    p = "cnn_stem."
    h = F.gelu(_bn(P, p + "stem.1", F.conv2d(x, P[p + "stem.0.weight"], P[p + "stem.0.bias"], stride=2, padding=1), train))
    h = F.gelu(_bn(P, p + "stage1.1", F.conv2d(h, P[p + "stage1.0.weight"], P[p + "stage1.0.bias"], stride=2, padding=1), train))
    f2 = _convnext(P, p + "stage1.3", h)
    f3 = _convnext(P, p + "stage2.2", _bn(P, p + "stage2.1", F.conv2d(f2, P[p + "stage2.0.weight"], P[p + "stage2.0.bias"]), train))
    f4 = _convnext(P, p + "stage3.2", _bn(P, p + "stage3.1", F.conv2d(f3, P[p + "stage3.0.weight"], P[p + "stage3.0.bias"]), train))
    return f2, f3, f4


STEM_V2_DROP_PATH = ((0.0, 0.0), (0.0, 0.1, 0.1), (0.1, 0.1))     # HQAViTv2_CIFAR100.py:772-773, :783-785, :797-798


def _spatial_ln(P, pre: str, x: Tensor) -> Tensor:
    """nn.LayerNorm([C, H, W], eps=1e-6) on an NCHW map, HQAViTv2_CIFAR100.py:766."""
    return F.layer_norm(x, tuple(x.shape[1:]), P[pre + ".weight"], P[pre + ".bias"], 1e-6)


def cnn_stem_v2(P, x: Tensor, train: bool, stem_drop: bool = True):
    """CNNStemModel.forward of the ConvNeXt-Tiny style stem, HQAViTv2_CIFAR100.py:809-829."""
    p = "cnn_stem."
    dps = STEM_V2_DROP_PATH if stem_drop else ((0.0, 0.0), (0.0, 0.0, 0.0), (0.0, 0.0))
    h = _spatial_ln(P, p + "stem.1", F.conv2d(x, P[p + "stem.0.weight"], P[p + "stem.0.bias"], stride=4))
    for i, dp in enumerate(dps[0]):
        h = _convnext(P, f"{p}stage2.{i}", h, dp, train)
    f2 = h
    h = F.conv2d(_spatial_ln(P, p + "downsample2.0", f2), P[p + "downsample2.1.weight"], P[p + "downsample2.1.bias"])
    for i, dp in enumerate(dps[1]):
        h = _convnext(P, f"{p}stage3.{i}", h, dp, train)
    f3 = h
    h = F.conv2d(_spatial_ln(P, p + "downsample3.0", f3), P[p + "downsample3.1.weight"], P[p + "downsample3.1.bias"])
    for i, dp in enumerate(dps[2]):
        h = _convnext(P, f"{p}stage4.{i}", h, dp, train)
    return f2, f3, h


def lmfa(P, pre: str, feat: Tensor, target_hw: int) -> Tensor:
    """LMFAdapter.forward, HQAViT_CIFAR100.py:819-849."""
    C = feat.shape[1]
    f1 = F.conv2d(feat, P[pre + ".dwconv_3x3.weight"], P[pre + ".dwconv_3x3.bias"], padding=1, groups=C)
    f2 = F.conv2d(feat, P[pre + ".dwconv_5x5.weight"], P[pre + ".dwconv_5x5.bias"], padding=2, groups=C)
    h = F.conv2d(torch.cat([f1, f2, feat], 1), P[pre + ".proj.weight"], P[pre + ".proj.bias"])
    if h.shape[2] != target_hw or h.shape[3] != target_hw:
        h = F.interpolate(h, size=(target_hw, target_hw), mode="bilinear", align_corners=False)
    return F.gelu(_ln(P, pre + ".norm", h.flatten(2).transpose(1, 2)))


def rrcv(P, pre: str, A: Tensor, H: int, W: int) -> Tensor:
    """RRCV.forward, HQAViT_CIFAR100.py:880-907 (rrcv_num_blocks ConvNeXt blocks)."""
    B, N, C = A.shape
    h = F.conv2d(A.permute(0, 2, 1).reshape(B, C, H, W), P[pre + ".reverse_proj.weight"], P[pre + ".reverse_proj.bias"])
    i = 0
    while f"{pre}.blocks.{i}.dwconv.weight" in P:
        h = _convnext(P, f"{pre}.blocks.{i}", h)
        i += 1
    h = F.conv2d(h, P[pre + ".reembed_proj.weight"], P[pre + ".reembed_proj.bias"])
    return A + P[pre + ".beta"] * _ln(P, pre + ".norm", h.flatten(2).transpose(1, 2))


def split_fusion(P, pre: str, T: Tensor, R: Tensor, train: bool) -> Tensor:
    """SplitFusion.forward, HQAViT_CIFAR100.py:941-965 (cat_mlp dropout is a fixed 0.1, :930)."""
    gate = torch.sigmoid(_lin(P, pre + ".gate_fc", _ln(P, pre + ".gate_norm", T + R)))
    t_add = T + gate * R
    h = F.gelu(_ln(P, pre + ".cat_mlp.1", _lin(P, pre + ".cat_mlp.0", torch.cat([T, R], -1))))
    t_cat = T + _drop(h, 0.1 if train else 0.0, train, pre + ".cat_mlp")
    w = F.softmax(P[pre + ".fusion_weights"], dim=0)
    return _ln(P, pre + ".final_norm", w[0] * t_add + w[1] * t_cat)


# --------------------------------------------------------------------------------------------------
# whole models
# --------------------------------------------------------------------------------------------------
def hqavit_stage_sizes(cfg):
    """[2,2,2,2] for depth 8 (HQAViT_CIFAR100.py:1189-1207), [2,2,6,2] for depth 12 (HQAViT_IN_Tiny.py:1398-1420)."""
    return (2, 2, cfg.depth - 6, 2)


def hqavit_forward(P, x: Tensor, cfg, train: bool = False, variant: str = "hqa", taps=None,
                   cat_dropout: Optional[bool] = None, sync=None, stem_drop: bool = True) -> Tensor:
    """HQAViT.forward, HQAViT_CIFAR100.py:1226-1277.  ``P`` is the model's state_dict (fp32 tensors);
    bank tensors are mutated in place in train mode, as the reference does.  A state dict with the ConvNeXt-Tiny
    style stem (``cnn_stem.downsample2.*``) takes HQAViTv2_CIFAR100.py's stem; the rest of that file's forward
    (:1262-1310) is the same code."""
    var = VARIANTS[variant]
    Hh = cfg.img_size // cfg.patch_size
    if "cnn_stem.downsample2.0.weight" in P:
        f2, f3, f4 = cnn_stem_v2(P, x, train, stem_drop)
    else:
        f2, f3, f4 = cnn_stem(P, x, train)
    R = {}
    for i, f in ((2, f2), (3, f3), (4, f4)):
        R[i] = rrcv(P, f"rrcv{i}", lmfa(P, f"lmfa{i}", f, Hh), Hh, Hh)
    T = patch_embed(P, x, cfg) + P["pos_embed"]
    T = _drop(T, cfg.dropout, train, "pos_drop")
    if taps is not None:
        taps["embed"] = T
    dpr = torch.linspace(0, cfg.drop_path, cfg.depth).tolist()
    blk = 0
    fuse_train = train if cat_dropout is None else cat_dropout
    for si, n in enumerate(hqavit_stage_sizes(cfg), start=1):
        if si >= 2:
            T = split_fusion(P, f"fuse{si}", T, R[si], fuse_train)
            if taps is not None:
                taps[f"fuse{si}"] = T
        for j in range(n):
            T = tl_block(P, f"stage{si}_blocks.{j}", T, cfg, var, train, dpr[blk], taps, sync)
            if taps is not None:
                taps[f"stage{si}_blocks.{j}"] = T
            blk += 1
    T = _ln(P, "norm", T).mean(1)
    return _lin(P, "head", T)


def qavit_forward(P, x: Tensor, cfg, train: bool = False, variant: str = "v1", taps=None, sync=None) -> Tensor:
    """QAViT.forward, QAViT.py:689-699 / QAViTv2.py:1045-1055."""
    var = VARIANTS[variant]
    T = patch_embed(P, x, cfg) + P["pos_embed"]
    T = _drop(T, cfg.dropout, train, "pos_drop")
    dpr = torch.linspace(0, cfg.drop_path, cfg.depth).tolist()
    for i in range(cfg.depth):
        T = quad_block(P, f"blocks.{i}", T, cfg, var, train, dpr[i], taps, sync)
        if taps is not None:
            taps[f"blocks.{i}"] = T
    return _lin(P, "head", _ln(P, "norm", T).mean(1))


def loss_fn(logits: Tensor, target: Tensor, label_smoothing: float = 0.12) -> Tensor:
    """nn.CrossEntropyLoss(label_smoothing=...), HQAViT_CIFAR100.py:1373."""
    return F.cross_entropy(logits, target, label_smoothing=label_smoothing)
