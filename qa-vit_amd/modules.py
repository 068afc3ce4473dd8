"""nn.Module mirror of the reference's model classes: same class names, constructor arguments, module tree
and ``state_dict`` keys, with every ``forward`` running on the HIP kernels of
libqavit_hip.so through ``functional``.  Parameters are ordinary fp32 ``nn.Parameter`` s held by stock
``nn.Linear`` / ``nn.LayerNorm`` / ``nn.Conv2d`` containers, so ``state_dict`` / ``load_state_dict`` /
``named_parameters`` / hooks / ``deepcopy`` (EMA) / any torch optimizer work unchanged; those containers'
own forwards are never used on the hot path.

Reference: HQAViT_CIFAR100.py:256-1138 (HQA blocks), QAViT.py:161-651 (v1), QAViTv2.py:460-1009 (v2).
"""
import math
from typing import Optional

import os

import torch
import torch.nn as nn

from . import functional as F
from . import kernels as K


def _hw(n: int) -> int:
    return int(math.sqrt(n))


def _drive(gen):
    """Run a step generator to its end -> its return value."""
    try:
        while True:
            next(gen)
    except StopIteration as e:
        return e.value


_UPMIX_SA = os.environ.get("QAVIT_UPMIX_SA", "1") != "0"   # the block's closing scale-add differentiated inside the up-mix backward launch
_LN_FAN = os.environ.get("QAVIT_LN_FAN", "1") != "0"     # norm1's five-way gradient fan-in summed inside its LayerNorm-backward launch
_BANK_PROJ2 = os.environ.get("QAVIT_BANK_PROJ2", "1") != "0"


class _Ctx:
    """Per-model knobs shared by all blocks: variant switches, data-parallel bank hook."""

    def __init__(self, variant: str):
        self.variant = variant
        self.ccf_norm = variant in ("hqa", "v2")
        self.dw_bias = variant in ("v1", "v2")
        self.dw_scale = variant in ("hqa", "v2")
        self.bank_mode = 1 if variant == "v1" else 0
        self.bank_sync = None          # callable(acc, local_batch) -> global batch (set by parallel.DataParallel)
        self.bank_writes = True
        # inside a QuadAttentionBlock.forward: the (k, v) copy the last bank write left beside the parameter (qavit_bank_apply snap_*),
        # i.e. what the next branch's backward must see -- saves that branch its own snapshot launch
        self.in_block = False
        self.in_model = False          # inside a model's forward (models.py): the copy survives from one block to the next
        self.snap = None


class GlobalTokenBank(nn.Module):
    """HQAViT_CIFAR100.py:275-321 (v1: QAViT.py:183-224, no update_count)."""

    def __init__(self, bank_size: int, embed_dim: int, with_counter: bool = True):
        super().__init__()
        self.bank_size, self.embed_dim = bank_size, embed_dim
        self.global_k = nn.Parameter(torch.randn(1, bank_size, embed_dim) * 0.02)
        self.global_v = nn.Parameter(torch.randn(1, bank_size, embed_dim) * 0.02)
        self.write_norm = nn.LayerNorm(embed_dim)
        self.write_compression = nn.Linear(embed_dim, embed_dim)
        self.write_gate = nn.Linear(embed_dim, bank_size)
        if with_counter:
            self.register_buffer("update_count", torch.tensor(0))

    def read(self, batch_size: int):
        return self.global_k.expand(batch_size, -1, -1), self.global_v.expand(batch_size, -1, -1)


class LinformerCompression(nn.Module):
    """HQAViT_CIFAR100.py:324-352; consumed inside the fused attention kernel (csrc/attn.hip)."""

    def __init__(self, seq_len: int, compressed_len: int):
        super().__init__()
        self.seq_len, self.compressed_len = seq_len, compressed_len
        self.E_k = nn.Parameter(torch.randn(seq_len, compressed_len) * 0.02)
        self.E_v = nn.Parameter(torch.randn(seq_len, compressed_len) * 0.02)


class _Branch(nn.Module):
    def __call__(self, *args, **kwargs):
        try:
            return super().__call__(*args, **kwargs)
        except BaseException:
            # a NaN rule deferred by the fused kernel (kernels.Runtime.pending_fix, per-device state) must not outlive the call that failed
            # between the deferral and the launch that would have consumed it: every later forward would raise "never consumed"
            K.Runtime.drop_pending_fix()
            raise

    def _writes(self) -> bool:
        """Will _write() write the bank from this branch's output?  (then the fused kernels leave their NaN rule to that launch)"""
        return bool(self.training and self._rt.bank_writes and hasattr(self, "norm"))

    def _write(self, out):
        rt = self._rt
        if self.training and rt.bank_writes and hasattr(self, "norm"):
            rt.snap = F.bank_write(out, self.norm.weight, self.norm.bias, self.global_bank, rt.bank_mode, rt.bank_sync,
                                   want_snap=rt.in_block and torch.is_grad_enabled())

    def _snap(self):
        """Forward-time copy of the bank rows if the previous branch of this block left one (else the consumer copies)."""
        rt = self._rt
        return rt.snap if rt.in_block else None


class EfficientSpatialWindowAttention(_Branch):
    """HQAViT_CIFAR100.py:403-469: window partition is a row-index table, not a copy."""

    def __init__(self, config, global_bank, rt: _Ctx):
        super().__init__()
        d = config.embed_dim
        self.config, self.global_bank, self._rt = config, global_bank, rt
        self.embed_dim, self.num_heads, self.head_dim = d, config.num_heads, d // config.num_heads
        self.window_size = config.window_size
        self.qkv = nn.Linear(d, 3 * d, bias=True)
        self.linformer = LinformerCompression(self.window_size ** 2, config.linformer_k)
        self.proj = nn.Linear(d, d)
        self.dropout = nn.Dropout(config.dropout)
        self.norm = nn.LayerNorm(d)
        self._site, self._site_attn = K.new_site(), K.new_site()

    def forward(self, x):
        B, N, C = x.shape
        Hs, ws = _hw(N), self.window_size
        if Hs % ws != 0:
            raise NotImplementedError("SWA window padding (HQAViT_CIFAR100.py:424-428) is not built yet: token grid "
                                      f"{Hs}x{Hs} is not a multiple of window {ws}")
        nw = Hs // ws
        tbl = None
        if nw > 1:
            def build():
                t = []
                for wy in range(nw):
                    for wx in range(nw):
                        for ty in range(ws):
                            for tx in range(ws):
                                t.append((wy * ws + ty) * Hs + wx * ws + tx)
                return t
            tbl = K.Runtime.get(x.device).table(("win", Hs, ws), build)
        p = self.dropout.p if self.training else 0.0
        if ((nw == 1 and N == 16) or (nw == 2 and N == 64)) and ws == 4 and self.linformer.seq_len == 16 and \
                F.branch_ok(0, x, 16, self.linformer.compressed_len, self.global_bank.bank_size, self.num_heads):
            # 16 tokens: one window = the image's tokens; 64 tokens: the four windows of the 8x8 grid, gathered in the kernel -- the whole
            # branch is one launch either way (csrc/branch_fwd.hip)
            out = F.BranchFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.linformer.E_k, self.linformer.E_v,
                                   self.global_bank.global_k, self.global_bank.global_v,
                                   dict(kind=0, attn_drop=(p, self._site_attn), proj_drop=(p, self._site), bank_snap=self._snap(), win_tbl=tbl,
                                        defer_fix=self._writes()))
            self._write(out)
            return out
        qkv = F.linear(x, self.qkv.weight, self.qkv.bias).reshape(B * N, 3 * C)
        spec = dict(mode=0, G=B * nw * nw, Nq=ws * ws, L=ws * ws, H=self.num_heads, D=self.head_dim,
                    KC=self.linformer.compressed_len, S=self.global_bank.bank_size, groups_per_b=nw * nw,
                    q_rows_per_b=N, k_rows_per_b=N, q_tbl=tbl, k_tbl=tbl, q_off=0, k_off=C, v_off=2 * C, q_rows=B * N)
        p = self.dropout.p if self.training else 0.0
        spec["drop"] = (p, self._site_attn)                    # efficient_attention(q, k, v, self.dropout.p, self.training), :461
        spec["bank_snap"] = self._snap()
        o = F.AttnFn.apply(qkv, None, self.linformer.E_k, self.linformer.E_v,
                           self.global_bank.global_k, self.global_bank.global_v, spec)
        out = F.linear(o.reshape(B, N, C), self.proj.weight, self.proj.bias, drop=(p, self._site))
        self._write(out)
        return out


class EfficientMultiScaleDilatedAttention(_Branch):
    """HQAViT_CIFAR100.py:472-532.  Only the K,V rows of ``qkv`` are applied to the pooled landmarks and
    only the Q rows to the full tokens (the reference computes and discards the rest, :505, :523)."""

    def __init__(self, config, global_bank, rt: _Ctx):
        super().__init__()
        d = config.embed_dim
        self.config, self.global_bank, self._rt = config, global_bank, rt
        self.embed_dim, self.num_heads, self.head_dim = d, config.num_heads, d // config.num_heads
        self.dilation_factors = config.dilation_factors
        self.qkv = nn.Linear(d, 3 * d, bias=True)
        self.linformer = LinformerCompression(128, config.linformer_k)
        self.landmark_pool = nn.AvgPool1d(config.landmark_pooling_stride, config.landmark_pooling_stride)
        self.proj = nn.Linear(d, d)
        self.dropout = nn.Dropout(config.dropout)
        self.norm = nn.LayerNorm(d)
        self._site, self._site_attn = K.new_site(), K.new_site()

    def forward(self, x, x_q=None):
        """``x_q``: optional second alias of the same tensor for the Q projection (lets the caller's fan-out node sum both
        gradient contributions in its one pass)."""
        x_q = x if x_q is None else x_q
        B, N, C = x.shape
        Hs = _hw(N)
        stride = self.config.landmark_pooling_stride

        def build():
            t = []
            for d in self.dilation_factors:
                for y in range(0, Hs, d):
                    for xx in range(0, Hs, d):
                        t.append(y * Hs + xx)
            return t[: (len(t) // stride) * stride]
        idx = K.Runtime.get(x.device).table(("msda", Hs, tuple(self.dilation_factors), stride), build)
        NP = idx.numel() // stride
        p = self.dropout.p if self.training else 0.0
        same = x_q is x or (x_q.data_ptr() == x.data_ptr() and x_q.shape == x.shape and x_q.stride() == x.stride())
        if same and NP <= (48 if N == 64 else 16) and NP <= self.linformer.seq_len and \
                F.branch_ok(1, x, NP, self.linformer.compressed_len, self.global_bank.bank_size, self.num_heads):
            out = F.BranchFn.apply(x, self.qkv.weight, self.qkv.bias, self.proj.weight, self.proj.bias, self.linformer.E_k, self.linformer.E_v,
                                   self.global_bank.global_k, self.global_bank.global_v,
                                   dict(kind=1, pool_idx=idx, pool_stride=stride, Lk=NP, attn_drop=(p, self._site_attn), proj_drop=(p, self._site),
                                        bank_snap=self._snap(), defer_fix=self._writes()))
            self._write(out)
            return out
        pooled = F.GatherPoolFn.apply(x, idx, stride)
        kv = F.linear(pooled, self.qkv.weight, self.qkv.bias, rows=(C, 2 * C)).reshape(B * NP, 2 * C)
        q = F.linear(x_q, self.qkv.weight, self.qkv.bias, rows=(0, C)).reshape(B * N, C)
        Lk = min(NP, self.linformer.seq_len)
        spec = dict(mode=0, G=B, Nq=N, L=Lk, H=self.num_heads, D=self.head_dim, KC=self.linformer.compressed_len,
                    S=self.global_bank.bank_size, groups_per_b=1, q_rows_per_b=N, k_rows_per_b=NP,
                    q_off=0, k_off=0, v_off=C, q_rows=B * N)
        p = self.dropout.p if self.training else 0.0
        spec["drop"] = (p, self._site_attn)                    # :524
        spec["bank_snap"] = self._snap()
        o = F.AttnFn.apply(q, kv, self.linformer.E_k, self.linformer.E_v,
                           self.global_bank.global_k, self.global_bank.global_v, spec)
        out = F.linear(o.reshape(B, N, C), self.proj.weight, self.proj.bias, drop=(p, self._site))
        self._write(out)
        return out


class EfficientChannelGroupAttention(_Branch):
    """HQAViT_CIFAR100.py:535-595: channel groups are rows of a [B*N*G, 32] view; q/k/v are one stacked GEMM."""

    def __init__(self, config, global_bank, rt: _Ctx):
        super().__init__()
        d = config.embed_dim
        self.config, self.global_bank, self._rt = config, global_bank, rt
        self.embed_dim, self.num_heads, self.head_dim = d, config.num_heads, d // config.num_heads
        self.num_groups = config.num_channel_groups
        self.channels_per_group = d // self.num_groups
        self.compress_c = d // 2
        self.compress_per_group = self.compress_c // self.num_groups
        cpg, ccg = self.channels_per_group, self.compress_per_group
        self.q_proj, self.k_proj, self.v_proj = nn.Linear(cpg, ccg), nn.Linear(cpg, ccg), nn.Linear(cpg, ccg)
        self.bank_k_proj, self.bank_v_proj = nn.Linear(d, ccg), nn.Linear(d, ccg)
        self.proj = nn.Linear(self.compress_c, d)
        self.dropout = nn.Dropout(config.dropout)
        self.norm = nn.LayerNorm(d)
        self._site, self._site_attn = K.new_site(), K.new_site()

    def forward(self, x):
        B, N, C = x.shape
        G, cpg, ccg, H = self.num_groups, self.channels_per_group, self.compress_per_group, self.num_heads
        bank = self.global_bank
        fused = F.cga_ok(x, G, H, bank.bank_size) and ccg == 16 and cpg == 32
        # .clone(): the reference's Linear flattens the EXPANDED bank, which copies -- its weight gradient sees the
        # forward-time bank, not the in-place writes that follow (HQAViT_CIFAR100.py:576-577)
        if _BANK_PROJ2 and x.is_cuda and bank.global_k.dtype == torch.float32:
            sh_k, sh_v = F.bank_proj2(bank, self._snap(), self.bank_k_proj, self.bank_v_proj)
        else:
            gk, gv = F.bank_snapshot(bank, self._snap())
            sh_k = F.linear(gk, self.bank_k_proj.weight, self.bank_k_proj.bias).reshape(bank.bank_size, ccg)
            sh_v = F.linear(gv, self.bank_v_proj.weight, self.bank_v_proj.bias).reshape(bank.bank_size, ccg)
        tbl = K.Runtime.get(x.device).table(("cga", N, G), lambda: [n * G + g for g in range(G) for n in range(N)])
        spec = dict(mode=1, G=B * G, Nq=N, L=N, H=H, D=ccg // H, S=bank.bank_size, groups_per_b=G,
                    q_rows_per_b=N * G, k_rows_per_b=N * G, q_tbl=tbl, k_tbl=tbl, q_off=0, k_off=ccg, v_off=2 * ccg,
                    q_rows=B * N * G)
        p = self.dropout.p if self.training else 0.0
        spec["drop"] = (p, self._site_attn)                    # :587
        if fused:
            out = F.CGABranchFn.apply(x, self.q_proj.weight, self.q_proj.bias, self.k_proj.weight, self.k_proj.bias, self.v_proj.weight, self.v_proj.bias,
                                      self.proj.weight, self.proj.bias, sh_k, sh_v,
                                      dict(G=G, H=H, spec=spec, attn_drop=(p, self._site_attn), proj_drop=(p, self._site), defer_fix=self._writes()))
            self._write(out)
            return out
        qkv = F.LinearStack3Fn.apply(x.reshape(B * N * G, cpg), self.q_proj.weight, self.q_proj.bias,
                                     self.k_proj.weight, self.k_proj.bias, self.v_proj.weight, self.v_proj.bias)
        o = F.AttnFn.apply(qkv, None, None, None, sh_k, sh_v, spec)
        out = F.linear(o.reshape(B, N, self.compress_c), self.proj.weight, self.proj.bias, drop=(p, self._site))
        self._write(out)
        return out


class CrossAttentionBranch(_Branch):
    """HQAViT_CIFAR100.py:598-626: K,V are projections of the (batch-invariant) bank, computed once."""

    def __init__(self, config, global_bank, rt: _Ctx):
        super().__init__()
        d = config.embed_dim
        self.config, self.global_bank, self._rt = config, global_bank, rt
        self.embed_dim, self.num_heads, self.head_dim = d, config.num_heads, d // config.num_heads
        self.q_proj, self.k_proj, self.v_proj, self.proj = nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d), nn.Linear(d, d)
        self.dropout = nn.Dropout(config.dropout)
        self._site, self._site_attn = K.new_site(), K.new_site()

    def forward(self, x):
        B, N, C = x.shape
        bank = self.global_bank
        if _BANK_PROJ2 and x.is_cuda and bank.global_k.dtype == torch.float32:
            sh_k, sh_v = F.bank_proj2(bank, self._snap(), self.k_proj, self.v_proj)                            # see CGA
        else:
            gk, gv = F.bank_snapshot(bank, self._snap())
            sh_k = F.linear(gk, self.k_proj.weight, self.k_proj.bias).reshape(bank.bank_size, C)
            sh_v = F.linear(gv, self.v_proj.weight, self.v_proj.bias).reshape(bank.bank_size, C)
        p = self.dropout.p if self.training else 0.0
        if F.branch_ok(2, x, 0, 0, bank.bank_size, self.num_heads):
            # inside a QuadAttentionBlock the next reader of this output is the compress-fuse node: it carries the NaN rule's rewrite
            # (qavit_cfuse_args.fix) -- unless a hook wants to see the output first
            defer = bool(F.DEFER_FIX_CFUSE and getattr(self._rt, "in_block", False) and getattr(self._rt, "cross_to_cfuse", False)
                         and not self._forward_hooks)
            return F.BranchFn.apply(x, self.q_proj.weight, self.q_proj.bias, self.proj.weight, self.proj.bias, None, None, sh_k, sh_v,
                                    dict(kind=2, attn_drop=(p, self._site_attn), proj_drop=(p, self._site), defer_fix=defer))
        q = F.linear(x, self.q_proj.weight, self.q_proj.bias).reshape(B * N, C)
        spec = dict(mode=1, G=B, Nq=N, L=0, H=self.num_heads, D=self.head_dim, S=bank.bank_size, q_off=0, k_off=0, v_off=0,
                    q_rows=B * N)
        p = self.dropout.p if self.training else 0.0
        spec["drop"] = (p, self._site_attn)                    # :624
        o = F.AttnFn.apply(q, None, None, None, sh_k, sh_v, spec)
        return F.linear(o.reshape(B, N, C), self.proj.weight, self.proj.bias, drop=(p, self._site))


class HybridFusion(nn.Module):
    def __init__(self, embed_dim, num_branches=4):
        super().__init__()
        self.fusion_weights = nn.Parameter(torch.ones(num_branches))


class BottleneckMLP(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, dropout=0.1):
        super().__init__()
        self.fc1 = nn.Linear(input_dim, hidden_dim)
        self.act = nn.GELU()
        self.dropout = nn.Dropout(dropout)
        self.fc2 = nn.Linear(hidden_dim, output_dim)
        self._s1, self._s2 = K.new_site(), K.new_site()


class DepthwiseConv2d(nn.Module):
    """HQAViT_CIFAR100.py:659-675 (scale, no bias) / QAViT.py:553-562 (bias, no scale) / QAViTv2.py:852-884 (both)."""

    def __init__(self, dim, kernel_size=3, bias=False, scale=True):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size, padding=kernel_size // 2, groups=dim, bias=bias)
        if scale:
            self.scale = nn.Parameter(torch.ones(1, dim, 1, 1) * 0.1)


class CCFFFN(nn.Module):
    """HQAViT_CIFAR100.py:678-712: fc1+GELU (GEMM epilogue) -> [LN -> dw3x3*scale -> LN] (one kernel) -> fc2."""

    def __init__(self, embed_dim, mlp_ratio=0.5, dropout=0.1, rt: Optional[_Ctx] = None):
        super().__init__()
        hidden = int(embed_dim * mlp_ratio)
        self._rt = rt
        self.fc1 = nn.Linear(embed_dim, hidden)
        self.act = nn.GELU()
        if rt.ccf_norm:
            self.dwconv_norm = nn.LayerNorm(hidden)
        self.dwconv = DepthwiseConv2d(hidden, 3, bias=rt.dw_bias, scale=rt.dw_scale)
        if rt.ccf_norm:
            self.post_dwconv_norm = nn.LayerNorm(hidden)
        self.fc2 = nn.Linear(hidden, embed_dim)
        self.dropout = nn.Dropout(dropout)
        if rt.ccf_norm:
            self.gamma = nn.Parameter(torch.ones(1) * 0.1)
        self._site = K.new_site()

    def branch(self, x, pre_norm: nn.LayerNorm):
        """-> (u, x_alias): u = dropout(fc2(mid(gelu(fc1(LN(x)))))); the caller applies gamma, drop-path and the residual, for
        which it uses ``x_alias`` (its gradient joins the LayerNorm's in one kernel)."""
        B, N, C = x.shape
        Hs = _hw(N)
        rt = self._rt
        h, xa = F.linear(x, self.fc1.weight, self.fc1.bias, ln=(pre_norm.weight, pre_norm.bias), eps=pre_norm.eps, act="gelu", alias=True)
        n1 = (self.dwconv_norm.weight, self.dwconv_norm.bias) if rt.ccf_norm else (None, None)
        n2 = (self.post_dwconv_norm.weight, self.post_dwconv_norm.bias) if rt.ccf_norm else (None, None)
        mid = F.ccf_mid(h, n1[0], n1[1], n2[0], n2[1], self.dwconv.dwconv.weight, self.dwconv.dwconv.bias,
                               self.dwconv.scale if rt.dw_scale else None, Hs, Hs, 1e-5)
        p = self.dropout.p if self.training else 0.0
        return F.linear(mid, self.fc2.weight, self.fc2.bias, drop=(p, self._site)), xa


class DropPath(nn.Module):
    def __init__(self, drop_prob=None):
        super().__init__()
        self.drop_prob = drop_prob


class QuadAttentionBlock(nn.Module):
    """HQAViT_CIFAR100.py:1037-1085."""

    def __init__(self, config, global_bank, drop_path=0.0, rt: Optional[_Ctx] = None):
        super().__init__()
        d = config.embed_dim
        self.config, self.embed_dim, self._rt = config, d, rt
        self.compressed_dim = d // config.compress_ratio
        self.norm1 = nn.LayerNorm(d)
        self.swa = EfficientSpatialWindowAttention(config, global_bank, rt)
        self.msda = EfficientMultiScaleDilatedAttention(config, global_bank, rt)
        self.cga = EfficientChannelGroupAttention(config, global_bank, rt)
        self.cross_attn = CrossAttentionBranch(config, global_bank, rt)
        for n in ("swa", "msda", "cga", "cross"):
            setattr(self, f"norm_{n}", nn.LayerNorm(d))
        for n in ("swa", "msda", "cga", "cross"):
            setattr(self, f"compress_{n}", nn.Linear(d, self.compressed_dim))
        self.fusion = HybridFusion(self.compressed_dim, 4)
        self.bottleneck_mlp = BottleneckMLP(4 * self.compressed_dim, d // config.bottleneck_ratio, d, config.dropout)
        self.drop_path1 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self.norm2 = nn.LayerNorm(d)
        self.ccf_ffn = CCFFFN(d, config.mlp_ratio, config.dropout, rt)
        self.drop_path2 = DropPath(drop_path) if drop_path > 0.0 else nn.Identity()
        self._dp = float(drop_path)
        self._dp1, self._dp2 = K.new_site(), K.new_site()

    def forward(self, x, tail=None):
        """``tail`` (a list): skip the closing x + droppath(gamma * u) and leave (x, u, gamma, drop-path spec) in it for the caller."""
        B, N, C = x.shape
        tr = self.training
        rt = self._rt
        rt.in_block = True
        if not rt.in_model:
            rt.snap = None
        try:
            return self._forward(x, B, N, C, tr, tail)
        except BaseException:
            K.Runtime.drop_pending_fix()                    # (see _Branch.__call__: the cross branch's rule waits for the compress-fuse launch)
            raise
        finally:
            rt.in_block = False
            if not rt.in_model:
                rt.snap = None

    def _forward(self, x, B, N, C, tr, tail=None):
        # norm1's output feeds four branches (MSDA twice); xr = x again, for the residual.  In backward the five gradients and the
        # residual's meet inside ONE LayerNorm-backward launch (summed on load), not in a k-way sum kernel in front of it
        if _LN_FAN and torch.is_grad_enabled() and x.requires_grad:
            *xns, xr = F.LayerNormFanFn.apply(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, 5)
        else:
            xn, xr = F.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps, alias=True)
            xns = F.FanOutFn.apply(xn, 5) if (torch.is_grad_enabled() and xn.requires_grad) else (xn,) * 5
        args = []
        rt = self._rt
        # the cross branch's output goes straight into the compress-fuse node below (no module hook in between reads it): that node's
        # launch may carry the branch's NaN rule
        rt.cross_to_cfuse = not any(m._forward_hooks or m._forward_pre_hooks for m in (self.norm_cross, self.compress_cross, self.fusion))
        try:
            for (name, branch), xb in zip((("swa", self.swa), ("msda", self.msda), ("cga", self.cga), ("cross", self.cross_attn)), xns):
                bo = branch(xb, xns[4]) if name == "msda" else branch(xb)
                nrm, cmp_ = getattr(self, f"norm_{name}"), getattr(self, f"compress_{name}")
                args += [bo, nrm.weight, nrm.bias, cmp_.weight, cmp_.bias]
        finally:
            rt.cross_to_cfuse = False
        fused = F.CompressFuseFn.apply(self.fusion.fusion_weights, self.norm_swa.eps, *args)
        mlp = self.bottleneck_mlp
        p = mlp.dropout.p if tr else 0.0
        dp = (self._dp if tr else 0.0)
        if F.mlp2_ok(fused, xr, mlp.fc1.weight, mlp.fc2.weight):
            # fc1 + GELU + dropout -> fc2 + dropout + drop path + residual: one launch each way (csrc/mlp2.hip)
            x = F.Mlp2Fn.apply(fused, xr, mlp.fc1.weight, mlp.fc1.bias, mlp.fc2.weight, mlp.fc2.bias,
                               dict(drop1=(p, mlp._s1), drop2=(p, mlp._s2), dp=(dp, self._dp1, N)))
        else:
            h = F.linear(fused, mlp.fc1.weight, mlp.fc1.bias, act="gelu", drop=(p, mlp._s1))
            x = F.linear(h, mlp.fc2.weight, mlp.fc2.bias, drop=(p, mlp._s2), dp=(dp, self._dp1, N), resid=xr)
        u, xr2 = self.ccf_ffn.branch(x, self.norm2)
        gamma = self.ccf_ffn.gamma if self._rt.ccf_norm else None
        if tail is not None:                                 # the caller applies the scale-add itself (TokenUpMix: functional.UpMixScaleAddFn)
            tail.extend((xr2, u, gamma, (dp, self._dp2, N)))
            return None
        return F.ScaleAddFn.apply(xr2, u, gamma, (dp, self._dp2, N))


class TokenLearner(nn.Module):
    """HQAViT_CIFAR100.py:971-1002."""

    def __init__(self, in_dim: int, num_out_tokens: int = 16):
        super().__init__()
        self.num_out_tokens = num_out_tokens
        self.attention = nn.Sequential(nn.LayerNorm(in_dim), nn.Linear(in_dim, num_out_tokens))

    def forward(self, x):
        ln, fc = self.attention[0], self.attention[1]
        if F.tl_ok(x, fc.weight) and not (ln._forward_hooks or fc._forward_hooks or ln._forward_pre_hooks or fc._forward_pre_hooks):
            return F.TokenLearnerFn.apply(x, ln.weight, ln.bias, fc.weight, fc.bias, ln.eps)      # one launch each way (csrc/tokens_tl.hip)
        scores, xa = F.linear(x, fc.weight, fc.bias, ln=(ln.weight, ln.bias), eps=ln.eps, alias=True)
        return F.TokMixFn.apply(scores, xa)


class TokenUpMix(nn.Module):
    """HQAViT_CIFAR100.py:1005-1031."""

    def __init__(self, embed_dim: int, num_in_tokens: int, num_out_tokens: int):
        super().__init__()
        self.num_in_tokens, self.num_out_tokens = num_in_tokens, num_out_tokens
        self.upsample_attn = nn.Linear(num_in_tokens, num_out_tokens)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, xc):
        return F.UpMixFn.apply(xc, self.upsample_attn.weight, self.upsample_attn.bias, self.norm.weight, self.norm.bias, self.norm.eps)


class QuadBlockWithTokenLearner(nn.Module):
    """HQAViT_CIFAR100.py:1091-1123 (no skip connection around the wrapped block)."""

    def __init__(self, config, global_bank, drop_path=0.0, use_token_learner=True, rt: Optional[_Ctx] = None):
        super().__init__()
        self.use_token_learner = use_token_learner
        if use_token_learner:
            M = config.num_learned_tokens
            sq = _hw(M)
            if sq * sq != M:                          # HQAViT_IN_Tiny.py:739-745
                M = max(4, sq * sq)
            self.token_learner = TokenLearner(config.embed_dim, M)
            self.token_upmix = TokenUpMix(config.embed_dim, M, (config.img_size // config.patch_size) ** 2)
        self.quad_block = QuadAttentionBlock(config, global_bank, drop_path, rt)

    def forward(self, x):
        if not self.use_token_learner:
            return self.quad_block(x)
        xc = self.token_learner(x)
        qb, up = self.quad_block, self.token_upmix
        if _UPMIX_SA and not (qb._forward_hooks or qb._forward_pre_hooks or up._forward_hooks or up._forward_pre_hooks):
            # the block's closing scale-add and the up-mix as one autograd node: one launch each way (functional.UpMixScaleAddFn), also
            # without gradients (evaluation: the forward's single launch)
            tail = []
            qb(xc, tail)
            xr2, u, gamma, dp = tail
            return F.UpMixScaleAddFn.apply(xr2, u, gamma, dp, up.upsample_attn.weight, up.upsample_attn.bias, up.norm.weight, up.norm.bias, up.norm.eps)
        return up(qb(xc))


class PatchProj(nn.Conv2d):
    """``patch_embed.proj``: a real nn.Conv2d (weights, hooks, Grad-CAM contract of test_hqa.py:239-259) whose
    forward is patch gather + MFMA GEMM; returns the [B,C,h,w] view the reference's conv would."""

    compute_dtype = torch.float32

    def forward(self, x):
        B, Cin, H, W = x.shape
        p = self.kernel_size[0]
        cols = F.patchify(x, p, self.compute_dtype)
        y = F.linear(cols, self.weight, self.bias)
        return y.view(B, H // p, W // p, self.out_channels).permute(0, 3, 1, 2)


class PatchEmbed(nn.Module):
    """HQAViT_CIFAR100.py:1129-1138."""

    def __init__(self, img_size=32, patch_size=4, in_channels=3, embed_dim=192):
        super().__init__()
        self.num_patches = (img_size // patch_size) ** 2
        self.patch_size = patch_size
        self.proj = PatchProj(in_channels, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.norm = nn.LayerNorm(embed_dim)

    def forward(self, x, pos: Optional[torch.Tensor] = None):
        t = self.proj(x).flatten(2).transpose(1, 2)
        return F.layer_norm(t, self.norm.weight, self.norm.bias, self.norm.eps, add=pos)


# ---------------------------------------------------------------------------------------------------
# CNN lateral path (second tier), channel-last on the HIP kernels throughout: depthwise convs =
# csrc/dwconv.hip, strided 3x3 = im2col + MFMA GEMM, 1x1 = GEMM, BatchNorm(+GELU) = csrc/bnorm.hip; no MIOpen call.
# ---------------------------------------------------------------------------------------------------
def _to_tokens(x):
    """[B,C,H,W] -> channel-last tokens [B, H*W, C] (one copy; everything downstream stays channel-last)."""
    B, C, H, W = x.shape
    return x.flatten(2).transpose(1, 2).contiguous()


def _conv1x1_tokens(t, conv: nn.Conv2d, alias: bool = False):
    """A 1x1 nn.Conv2d applied to channel-last tokens is a GEMM over the channel axis (weight [Cout,Cin,1,1]).
    ``alias`` -> (y, t_alias): hand ``t_alias`` to t's other consumer (its gradient is added in this GEMM's backward epilogue)."""
    return F.linear(t, conv.weight, conv.bias, alias=alias)


def _bn_tokens(t, bn: nn.BatchNorm2d, training: bool, gelu: bool = False, bump: bool = True):
    """nn.BatchNorm2d (+ nn.GELU when ``gelu``) on channel-last tokens == batch norm over the rows of [B*H*W, C]
    (per-rank batch statistics) -- csrc/bnorm.hip; anything it does not cover raises (no stock-op fallback).
    ``bump=False``: the caller has counted this batch in ``num_batches_tracked`` already (one launch for all its BatchNorms)."""
    B, N, C = t.shape
    if bump and training and bn.num_batches_tracked is not None:
        bn.num_batches_tracked.add_(1)
    if not (K.bn_supported(t.dtype, C) and bn.running_mean is not None and bn.momentum is not None and bn.affine):
        # no stock-op fallback on the hot path: the reference's stems are nn.BatchNorm2d(32 | 64 | 128 | 256) with affine parameters, running
        # statistics and a fixed momentum (HQAViT_CIFAR100.py:753-775); anything else has no HIP kernel and says so
        raise NotImplementedError(f"BatchNorm over {C} channels ({t.dtype}; affine={bn.affine}, track_running_stats={bn.running_mean is not None}, "
                                  f"momentum={bn.momentum}) has no HIP kernel: csrc/bnorm.hip covers channel counts its 16-byte vectors tile, "
                                  "affine, with running statistics and a fixed momentum")
    y = F.BatchNormFn.apply(t.reshape(B * N, C), bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.momentum, bn.eps, gelu, training)
    return y.reshape(B, N, C)


class ConvNeXtBlock(nn.Module):
    """HQAViT_CIFAR100.py:718-739 on channel-last tokens: dw7x7 (csrc/dwconv.hip) -> LN+Linear+GELU -> Linear+residual.
    ``layer_scale_init_value`` adds the v2 file's per-channel layer scale ``gamma`` and its drop path
    (HQAViTv2_CIFAR100.py:718-750); ``None`` keeps the v1 class (no ``gamma`` key in the state_dict)."""

    def __init__(self, dim, drop_path=0.0, layer_scale_init_value=None):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = nn.LayerNorm(dim, eps=1e-6)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.act = nn.GELU()
        self.pwconv2 = nn.Linear(4 * dim, dim)
        self.layer_scale = layer_scale_init_value is not None
        if self.layer_scale:
            self.gamma = nn.Parameter(layer_scale_init_value * torch.ones(dim)) if layer_scale_init_value > 0 else None
        self.drop_path = DropPath(drop_path) if (self.layer_scale and drop_path > 0.0) else nn.Identity()
        self._site = K.new_site()

    def forward_tokens(self, t, H, W):
        if torch.is_grad_enabled() and t.requires_grad:      # t also feeds the residual: its gradient is added inside the dwconv backward
            h, t = F.DwConvFn.apply(t, self.dwconv.weight, self.dwconv.bias, H, W, True)
        else:
            h = F.DwConvFn.apply(t, self.dwconv.weight, self.dwconv.bias, H, W)
        h = F.linear(h, self.pwconv1.weight, self.pwconv1.bias, ln=(self.norm.weight, self.norm.bias), eps=self.norm.eps, act="gelu")
        p = self.drop_path.drop_prob if (self.training and isinstance(self.drop_path, DropPath) and self.drop_path.drop_prob) else 0.0
        if self.layer_scale and self.gamma is not None:
            u = F.linear(h, self.pwconv2.weight, self.pwconv2.bias)
            return F.ChanScaleAddFn.apply(t, u, self.gamma, (p, self._site, H * W))
        return F.linear(h, self.pwconv2.weight, self.pwconv2.bias, resid=t, dp=(p, self._site, H * W) if p > 0.0 else None)

    def forward(self, x):                                   # NCHW surface of the reference class
        B, C, H, W = x.shape
        return self.forward_tokens(_to_tokens(x), H, W).transpose(1, 2).reshape(B, C, H, W)


_STEM3_STREAM = os.environ.get("QAVIT_STEM3_STREAM", "0") != "0"      # diagnostic only: see CNNStemModel.forward_tokens_scales
_STEM3 = {}


def _stem3_stream(device):
    key = torch.device(device).index or 0
    st = _STEM3.get(key)
    if st is None:
        st = _STEM3[key] = torch.cuda.Stream(device=device)
    return st


class CNNStemModel(nn.Module):
    """HQAViT_CIFAR100.py:742-793, channel-last throughout: the two strided 3x3 convolutions are im2col + MFMA GEMM,
    the 1x1 convolutions are GEMMs, the depthwise 7x7 is csrc/dwconv.hip, BatchNorm(+GELU) is csrc/bnorm.hip."""

    def __init__(self, in_ch=3, c2=64, c3=128, c4=256, norm_layer=nn.LayerNorm):
        super().__init__()
        self.stem = nn.Sequential(nn.Conv2d(in_ch, 32, 3, stride=2, padding=1), nn.BatchNorm2d(32), nn.GELU())
        self.stage1 = nn.Sequential(nn.Conv2d(32, c2, 3, stride=2, padding=1), nn.BatchNorm2d(c2), nn.GELU(), ConvNeXtBlock(c2))
        self.stage2 = nn.Sequential(nn.Conv2d(c2, c3, 1), nn.BatchNorm2d(c3), ConvNeXtBlock(c3))
        self.stage3 = nn.Sequential(nn.Conv2d(c3, c4, 1), nn.BatchNorm2d(c4), ConvNeXtBlock(c4))

    def _conv3x3s2_tokens(self, src, conv: nn.Conv2d, bn: nn.BatchNorm2d, dims, cdt):
        """conv3x3/s2/p1 as im2col + MFMA GEMM, then BatchNorm + GELU, channel-last in and out."""
        B = dims[0]
        cols = F.Im2ColFn.apply(src, dims, cdt)
        t = F.linear(cols, conv.weight, conv.bias, xpad=cols.shape[1] != conv.weight[0].numel()).reshape(B, -1, conv.out_channels)
        return _bn_tokens(t, bn, self.training, gelu=True, bump=False)

    def forward_tokens(self, x, cdt):
        """-> (F2, F3, F4) as channel-last tokens [B, h*w, c] in the compute dtype, and (h, w)."""
        return _drive(self.forward_tokens_steps(x, cdt))

    def forward_tokens_steps(self, x, cdt):
        """``forward_tokens`` as a generator that yields between its stages (the model interleaves this chain's launches with the
        token path's: models.HQAViT.forward); the result is the generator's return value."""
        B, Cin, H, W = x.shape
        if self.training:
            # the four BatchNorms' num_batches_tracked in ONE launch (four 1-element adds sat between the kernels of the forward's first chain)
            nbt = [bn.num_batches_tracked for bn in (self.stem[1], self.stage1[1], self.stage2[1], self.stage3[1]) if bn.num_batches_tracked is not None]
            if nbt:
                torch._foreach_add_(nbt, 1)
        with torch.autocast("cuda", enabled=False):
            t = F.stamp(self._conv3x3s2_tokens(x, self.stem[0], self.stem[1], (B, Cin, H, W, 3, 2, 1), cdt), "lat.stem0")
            yield
            H1, W1 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            t = self._conv3x3s2_tokens(t, self.stage1[0], self.stage1[1], (B, self.stem[0].out_channels, H1, W1, 3, 2, 1), cdt)
            yield
            h, w = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
            f2 = self.stage1[3].forward_tokens(t, h, w)
            yield
            # f2 / f3 also leave as lateral features: the aliases carry those consumers' gradients into the 1x1 convs' backward GEMMs
            t, f2 = _conv1x1_tokens(f2, self.stage2[0], alias=True)
            f3 = self.stage2[2].forward_tokens(_bn_tokens(t, self.stage2[1], self.training, bump=False), h, w)
            yield
            t, f3 = _conv1x1_tokens(f3, self.stage3[0], alias=True)
            f4 = self.stage3[2].forward_tokens(_bn_tokens(t, self.stage3[1], self.training, bump=False), h, w)
        return (f2, f3, f4), (h, w)

    def forward_tokens_scales(self, x, cdt):
        """``forward_tokens`` one lateral feature at a time: a generator that yields (F2, (h, w)), then F3, then F4, running each stem
        stage only when its feature is asked for.  The model asks for scale i + 1 AFTER it has built the token path's stage i
        (models.HQAViT: QAVIT_LATERAL_ORDER 3), so the autograd nodes of stem stage i + 1 are younger than that stage's blocks and
        the engine -- which runs the youngest ready node first -- takes their backward right after LMFAdapter i + 1's, beside the
        token path's stage-i backward, instead of at the very end of the pass where every stem node used to queue up."""
        B, Cin, H, W = x.shape
        if self.training:
            nbt = [bn.num_batches_tracked for bn in (self.stem[1], self.stage1[1], self.stage2[1], self.stage3[1]) if bn.num_batches_tracked is not None]
            if nbt:
                torch._foreach_add_(nbt, 1)
        with torch.autocast("cuda", enabled=False):
            t = F.stamp(self._conv3x3s2_tokens(x, self.stem[0], self.stem[1], (B, Cin, H, W, 3, 2, 1), cdt), "lat.stem0")
            H1, W1 = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
            t = self._conv3x3s2_tokens(t, self.stage1[0], self.stage1[1], (B, self.stem[0].out_channels, H1, W1, 3, 2, 1), cdt)
            h, w = (H1 + 2 - 3) // 2 + 1, (W1 + 2 - 3) // 2 + 1
            f2 = self.stage1[3].forward_tokens(t, h, w)
            t, f2 = _conv1x1_tokens(f2, self.stage2[0], alias=True)      # (the alias carries F2's lateral gradient into this conv's backward GEMM)
        yield f2, (h, w)
        with torch.autocast("cuda", enabled=False):
            f3 = self.stage2[2].forward_tokens(_bn_tokens(t, self.stage2[1], self.training, bump=False), h, w)
            t, f3 = _conv1x1_tokens(f3, self.stage3[0], alias=True)
        yield f3
        with torch.autocast("cuda", enabled=False):
            s3 = _stem3_stream(x.device) if (_STEM3_STREAM and x.is_cuda and torch.is_grad_enabled() and t.requires_grad) else None
            if s3 is None:
                f4 = self.stage3[2].forward_tokens(_bn_tokens(t, self.stage3[1], self.training, bump=False), h, w)
            else:
                # DIAGNOSTIC (QAVIT_STEM3_STREAM=1, off by default): the stem's last stage on a THIRD stream (so that its backward, replayed
                # on the forward stream, is a graph branch of its own).  Eager it runs; under hipGraph capture of the training step the
                # process dies of a segmentation fault INSIDE hipStreamEndCapture (python -X faulthandler: torch/cuda/graphs.py capture_end),
                # also when the stream is joined explicitly before the capture ends -- the reproducer of the "three-stream capture crash"
                # DESIGN.md section 6 carries since round 2.
                cur = torch.cuda.current_stream(x.device)
                s3.wait_stream(cur)
                with torch.cuda.stream(s3):
                    f4 = self.stage3[2].forward_tokens(_bn_tokens(t, self.stage3[1], self.training, bump=False), h, w)
                cur.wait_stream(s3)
                t.record_stream(s3)
                f4.record_stream(cur)
        yield f4

    def forward(self, x):                                   # NCHW surface of the reference class
        (f2, f3, f4), (h, w) = self.forward_tokens(x, x.dtype)
        B = x.shape[0]
        return tuple(f.transpose(1, 2).reshape(B, -1, h, w) for f in (f2, f3, f4))


def _spatial_ln_tokens(t, ln: nn.LayerNorm):
    """nn.LayerNorm([C,H,W]) on channel-last tokens (csrc/spatial_ln.hip): a sample is one row of N*C elements."""
    B, N, C = t.shape
    if not K.spatial_ln_supported(N, C):
        raise RuntimeError(f"spatial LayerNorm over [{C},{N}] has no HIP kernel (N*C must be 4096, 8192 or 16384)")
    return F.SpatialLayerNormFn.apply(t, ln.weight, ln.bias, ln.eps)


class CNNStemModelV2(nn.Module):
    """HQAViTv2_CIFAR100.py:753-829 (ConvNeXt-Tiny style): 4x4/s4 patchify conv + LayerNorm([c2,8,8]); [2,3,2] ConvNeXt
    blocks with layer scale; LayerNorm([c,8,8]) + 1x1 conv between the stages.  Channel-last throughout: the patchify
    conv is patch gather + MFMA GEMM, the 1x1 convolutions are GEMMs."""

    def __init__(self, in_ch=3, c2=64, c3=128, c4=256, norm_layer=nn.LayerNorm, hw=8):
        super().__init__()
        ls = 1e-6
        self.stem = nn.Sequential(nn.Conv2d(in_ch, c2, kernel_size=4, stride=4), nn.LayerNorm([c2, hw, hw], eps=1e-6))
        self.stage2 = nn.Sequential(ConvNeXtBlock(c2, 0.0, ls), ConvNeXtBlock(c2, 0.0, ls))
        self.downsample2 = nn.Sequential(nn.LayerNorm([c2, hw, hw], eps=1e-6), nn.Conv2d(c2, c3, kernel_size=1))
        self.stage3 = nn.Sequential(ConvNeXtBlock(c3, 0.0, ls), ConvNeXtBlock(c3, 0.1, ls), ConvNeXtBlock(c3, 0.1, ls))
        self.downsample3 = nn.Sequential(nn.LayerNorm([c3, hw, hw], eps=1e-6), nn.Conv2d(c3, c4, kernel_size=1))
        self.stage4 = nn.Sequential(ConvNeXtBlock(c4, 0.1, ls), ConvNeXtBlock(c4, 0.1, ls))
        self.apply(self._init_weights)

    @staticmethod
    def _init_weights(m):
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            nn.init.trunc_normal_(m.weight, std=0.02)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)

    def forward_tokens(self, x, cdt):
        return _drive(self.forward_tokens_steps(x, cdt))

    def forward_tokens_steps(self, x, cdt):
        B, Cin, H, W = x.shape
        with torch.autocast("cuda", enabled=False):
            conv = self.stem[0]
            p = conv.kernel_size[0]
            h, w = H // p, W // p
            t = F.linear(F.patchify(x, p, cdt), conv.weight, conv.bias).reshape(B, h * w, conv.out_channels)
            t = _spatial_ln_tokens(t, self.stem[1])
            for blk in self.stage2:
                t = blk.forward_tokens(t, h, w)
                yield
            f2 = t
            t = _conv1x1_tokens(_spatial_ln_tokens(f2, self.downsample2[0]), self.downsample2[1])
            for blk in self.stage3:
                t = blk.forward_tokens(t, h, w)
                yield
            f3 = t
            t = _conv1x1_tokens(_spatial_ln_tokens(f3, self.downsample3[0]), self.downsample3[1])
            for blk in self.stage4:
                t = blk.forward_tokens(t, h, w)
                yield
        return (f2, f3, t), (h, w)

    def forward(self, x):                                   # NCHW surface of the reference class
        (f2, f3, f4), (h, w) = self.forward_tokens(x, x.dtype)
        B = x.shape[0]
        return tuple(f.transpose(1, 2).reshape(B, -1, h, w) for f in (f2, f3, f4))


_LMF_GATHER = os.environ.get("QAVIT_LMF_GATHER", "1") != "0"      # 0: separate depthwise convolutions + torch.cat


class LMFAdapter(nn.Module):
    """HQAViT_CIFAR100.py:799-849 on channel-last tokens."""

    def __init__(self, in_channels: int, embed_dim: int, target_hw: int = 8):
        super().__init__()
        self.in_channels, self.embed_dim, self.target_hw = in_channels, embed_dim, target_hw
        self.dwconv_3x3 = nn.Conv2d(in_channels, in_channels, 3, padding=1, groups=in_channels)
        self.dwconv_5x5 = nn.Conv2d(in_channels, in_channels, 5, padding=2, groups=in_channels)
        self.proj = nn.Conv2d(3 * in_channels, embed_dim, 1)
        self.norm = nn.LayerNorm(embed_dim)
        self.act = nn.GELU()

    def forward_tokens(self, t, H, W):
        if _LMF_GATHER and H % 8 == 0 and W % 8 == 0:
            cat = F.LmfGatherFn.apply(t, self.dwconv_3x3.weight, self.dwconv_3x3.bias, self.dwconv_5x5.weight, self.dwconv_5x5.bias, H, W)
        else:
            f1 = F.DwConvFn.apply(t, self.dwconv_3x3.weight, self.dwconv_3x3.bias, H, W)
            f2 = F.DwConvFn.apply(t, self.dwconv_5x5.weight, self.dwconv_5x5.bias, H, W)
            cat = torch.cat([f1, f2, t], -1)
        h = _conv1x1_tokens(cat, self.proj)
        if H != self.target_hw or W != self.target_hw:
            # HQAViT_CIFAR100.py:840-842 resizes bilinearly when the stem's map is not the token grid.  Every shipped configuration has
            # stem stride 4 = patch size 4 (8x8 on 8x8, 16x16 on 16x16): there is no HIP kernel for the resize and no stock-op fallback
            raise NotImplementedError(f"LMFAdapter: stem map {H}x{W} != token grid {self.target_hw}x{self.target_hw}: the bilinear resize "
                                      "(HQAViT_CIFAR100.py:840-842) is not built -- no shipped configuration reaches it")
        return F.layer_norm(h, self.norm.weight, self.norm.bias, self.norm.eps, act="gelu")

    def forward(self, feat):
        B, C, H, W = feat.shape
        return self.forward_tokens(_to_tokens(feat), H, W)


class RRCV(nn.Module):
    """HQAViT_CIFAR100.py:855-907 on channel-last tokens (the reverse / re-embed 1x1 convolutions are GEMMs)."""

    def __init__(self, embed_dim: int, rec_channels: int = 64, num_blocks: int = 1, layer_scale_init_value=None):
        super().__init__()
        self.embed_dim, self.rec_channels = embed_dim, rec_channels
        self.reverse_proj = nn.Conv2d(embed_dim, rec_channels, 1)
        self.blocks = nn.ModuleList([ConvNeXtBlock(rec_channels, 0.0, layer_scale_init_value) for _ in range(num_blocks)])
        self.reembed_proj = nn.Conv2d(rec_channels, embed_dim, 1)
        self.norm = nn.LayerNorm(embed_dim)
        self.beta = nn.Parameter(torch.tensor(0.1))

    def forward(self, A, H: int, W: int):
        h, A = _conv1x1_tokens(A, self.reverse_proj, alias=True)     # A also feeds the residual below
        for blk in self.blocks:
            h = blk.forward_tokens(h, H, W)
        t = _conv1x1_tokens(h, self.reembed_proj)
        # own kernel: the stock `beta * x` backward reduces with a memset-initialised semaphore buffer (see Mix2Fn)
        return F.ScaleAddFn.apply(A, F.layer_norm(t, self.norm.weight, self.norm.bias, self.norm.eps), self.beta, (0.0, 0, 1))


_MIX3 = os.environ.get("QAVIT_MIX3", "1") != "0"              # 0: SplitFusion's dropout, add and blend as three launches


class SplitFusion(nn.Module):
    """HQAViT_CIFAR100.py:913-965."""

    def __init__(self, embed_dim: int, use_learnable_weights: bool = True):
        super().__init__()
        self.embed_dim = embed_dim
        self.gate_norm = nn.LayerNorm(embed_dim)
        self.gate_fc = nn.Linear(embed_dim, embed_dim)
        self.cat_mlp = nn.Sequential(nn.Linear(2 * embed_dim, embed_dim), nn.LayerNorm(embed_dim), nn.GELU(), nn.Dropout(0.1))
        if use_learnable_weights:
            self.fusion_weights = nn.Parameter(torch.tensor([0.75, 0.25]))
        else:
            self.register_buffer("fusion_weights", torch.tensor([0.75, 0.25]))
        self.final_norm = nn.LayerNorm(embed_dim)
        self._site = K.new_site()

    def forward(self, T, R):
        gn, gf = self.gate_norm, self.gate_fc
        grad = torch.is_grad_enabled()
        fn = self.final_norm
        own = isinstance(self.fusion_weights, nn.Parameter) and self.fusion_weights.numel() == 2 and (T.numel() * T.element_size()) % 16 == 0
        # gate + blend + final norm as ONE node (functional.GateMix3LayerNormFn): T then feeds three expressions, not four (its gradient from
        # the gate's pass-through and from the blend leave that node as one tensor)
        tail = (own and _MIX3 and F.MIX3_LN and F.GATE_MIX3_LN and T.dtype == R.dtype and T.numel() < 2 ** 32
                and not (fn._forward_hooks or fn._forward_pre_hooks))
        # T feeds four (three) expressions and R three: one k-way gradient sum each instead of autograd's pairwise adds
        if tail:
            T0, T2, T3 = F.FanOutFn.apply(T, 3) if (grad and T.requires_grad) else (T,) * 3
            T1 = T3
        else:
            T0, T1, T2, T3 = F.FanOutFn.apply(T, 4) if (grad and T.requires_grad) else (T,) * 4
        R0, R1, R2 = F.FanOutFn.apply(R, 3) if (grad and R.requires_grad) else (R,) * 3
        gl = F.linear(T0 + R0, gf.weight, gf.bias, ln=(gn.weight, gn.bias), eps=gn.eps)
        c0, c1 = self.cat_mlp[0], self.cat_mlp[1]
        # Linear(2C -> C) on cat([T, R]) = T W[:, :C]^T + R W[:, C:]^T + b: two accumulating GEMMs, no 2C-wide cat buffer (and no
        # slice copies of its gradient)
        Cc = T2.shape[-1]
        if F.linear_cat_ok(T2, R2, c0.weight) and not (c0._forward_hooks or c0._forward_pre_hooks):
            h = F.LinearCatFn.apply(T2, R2, c0.weight, c0.bias)      # ... and in bf16 ONE GEMM whose A operand switches source at column C
        else:
            h = F.linear(T2, c0.weight, c0.bias, cols=(0, Cc))
            h = F.linear(R2, c0.weight, None, cols=(Cc, Cc), resid=h)
        h = F.layer_norm(h, c1.weight, c1.bias, c1.eps, act="gelu")
        drop = (self.cat_mlp[3].p if self.training else 0.0, self._site)
        if tail and gl.dtype == h.dtype == T3.dtype:
            t3, r1, g1, hh = T3.contiguous(), R1.contiguous(), gl.contiguous(), h.contiguous()
            if K.mix3_ln_ok(t3, r1, hh, Cc) and K.mix3_ln_ok(t3, g1, hh, Cc):
                return F.GateMix3LayerNormFn.apply(t3, r1, g1, hh, self.fusion_weights, drop, fn.weight, fn.bias, fn.eps)
        if (T1.numel() * T1.element_size()) % 16 == 0 and T1.dtype == R1.dtype == gl.dtype:
            t_add = F.GateMixFn.apply(T1, R1, gl)            # T + sigmoid(gate) * R in one kernel
        else:
            t_add = T1 + torch.sigmoid(gl) * R1
        if own and _MIX3 and T3.dtype == h.dtype == t_add.dtype and t_add.numel() < 2 ** 32:
            ta, t3, hh = t_add.contiguous(), T3.contiguous(), h.contiguous()
            if F.MIX3_LN and not (fn._forward_hooks or fn._forward_pre_hooks) and K.mix3_ln_ok(ta, t3, hh, ta.shape[-1]):
                # the blend and the final norm: one launch each way (functional.Mix3LayerNormFn)
                return F.Mix3LayerNormFn.apply(ta, t3, hh, self.fusion_weights, drop, fn.weight, fn.bias, fn.eps)
            # dropout, the add and the blend in one kernel each way
            mixed = F.Mix3Fn.apply(t_add, T3, h, self.fusion_weights, drop)
            return F.layer_norm(mixed, fn.weight, fn.bias, fn.eps)
        h = F.dropout(h, self.cat_mlp[3].p, self._site, self.training)
        if own:
            mixed = F.Mix2Fn.apply(t_add, T3 + h, self.fusion_weights)
        else:
            w = torch.softmax(self.fusion_weights, 0).to(T.dtype)
            mixed = w[0] * t_add + w[1] * (T3 + h)
        return F.layer_norm(mixed, fn.weight, fn.bias, fn.eps)
