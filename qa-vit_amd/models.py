"""Top-level models: ``HQAViT`` (HQAViT_CIFAR100.py:1141-1277, HQAViT_IN_Tiny.py:1362-1494) and ``QAViT``
(QAViT.py:654-699, QAViTv2.py:1011-1055).  ``Model(config)``, ``forward(x[B,3,H,W]) -> logits``,
``state_dict`` layout, attribute surface (``patch_embed.proj``, ``pos_embed``, ``head``, ``global_bank``,
``cnn_stem`` / ``lmfa*`` / ``rrcv*`` / ``fuse*``, ``blocks``) follow the reference; the arithmetic runs in
libqavit_hip.so.

Compute dtype: under ``torch.autocast(dtype=bf16)`` (what the reference's training loops use,
HQAViT_CIFAR100.py:1401-1403) activations are bf16 with fp32 accumulation; otherwise fp32 (exact-fp32 MFMA),
which is the parity path.  ``model.compute_dtype = torch.bfloat16`` forces bf16 without autocast.
"""
import contextlib
import os

import torch
import torch.nn as nn

from . import functional as F
from . import kernels as K
from . import modules as M
from .config import HQAViTConfig, QAViTConfig  # noqa: F401


def _init_weights(m):
    """HQAViT._init_weights, HQAViT_CIFAR100.py:1215-1224."""
    if isinstance(m, nn.Linear):
        nn.init.trunc_normal_(m.weight, std=0.02)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.LayerNorm):
        nn.init.constant_(m.bias, 0)
        nn.init.constant_(m.weight, 1.0)
    elif isinstance(m, nn.Conv2d):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class _Base(nn.Module):
    compute_dtype = None
    _sync_reducer = None          # parallel.GradReducer when data-parallel (see parallel.SyncPoint)

    def _sync(self, T, tag):
        if self._sync_reducer is not None and torch.is_grad_enabled():
            from .parallel import SyncPoint
            return SyncPoint.apply(T, self._sync_reducer, tag)
        return T

    def _dtype(self, x):
        if self.compute_dtype is not None:
            return self.compute_dtype
        if torch.is_autocast_enabled():
            return torch.get_autocast_dtype("cuda")
        return torch.float32

    def _check(self, x):
        if not x.is_cuda:
            raise RuntimeError("qavit_amd models run on the GPU only (HIP kernels; there is no CPU fallback). "
                               "Move the model and the input to cuda.")

    def set_bank_sync(self, fn):
        """Data-parallel hook: ``fn(acc[S*C], local_batch) -> global_batch`` all-reduces the bank statistics."""
        self._rt.bank_sync = fn

    def _head(self, T):
        T = F.layer_norm(T, self.norm.weight, self.norm.bias, self.norm.eps)
        return F.linear(F.TokenMeanFn.apply(T), self.head.weight, self.head.bias)


_LATERAL_STREAM = os.environ.get("QAVIT_LATERAL_STREAM", "1") != "0"
_LATERAL_ORDER = int(os.environ.get("QAVIT_LATERAL_ORDER", "4"))
_EARLY_FLUSH = os.environ.get("QAVIT_EARLY_FLUSH", "0") != "0"   # deferred weight-gradient work of the token path launched when ITS backward ends
_DONE = object()
_SIDE = {}


def _side_stream(device):
    key = torch.device(device).index or 0
    st = _SIDE.get(key)
    if st is None:
        st = _SIDE[key] = torch.cuda.Stream(device=device)
    return st


class HQAViT(_Base):
    """``stem="v1"``: HQAViT_CIFAR100.py / HQAViT_IN_Tiny.py (BatchNorm stem).  ``stem="v2"``: HQAViTv2_CIFAR100.py -- the
    ConvNeXt-Tiny style stem and layer-scaled ConvNeXt blocks (also inside RRCV); everything else is identical."""

    def __init__(self, config: HQAViTConfig, variant: str = "hqa", stem: str = "v1"):
        super().__init__()
        if stem not in ("v1", "v2"):
            raise ValueError("stem must be 'v1' or 'v2'")
        self.config = config
        self._rt = M._Ctx(variant)
        self.num_patches = (config.img_size // config.patch_size) ** 2
        self.H = self.W = config.img_size // config.patch_size
        d = config.embed_dim
        self.patch_embed = M.PatchEmbed(config.img_size, config.patch_size, config.in_channels, d)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, d))
        self.pos_drop = nn.Dropout(config.dropout)
        self.global_bank = M.GlobalTokenBank(config.global_bank_size, d)
        if stem == "v2":
            self.cnn_stem = M.CNNStemModelV2(config.in_channels, config.cnn_c2, config.cnn_c3, config.cnn_c4, hw=config.img_size // 4)
        else:
            self.cnn_stem = M.CNNStemModel(config.in_channels, config.cnn_c2, config.cnn_c3, config.cnn_c4)
        for i, c in ((2, config.cnn_c2), (3, config.cnn_c3), (4, config.cnn_c4)):
            setattr(self, f"lmfa{i}", M.LMFAdapter(c, d, target_hw=self.H))
        for i in (2, 3, 4):
            setattr(self, f"rrcv{i}", M.RRCV(d, config.rrcv_channels, config.rrcv_num_blocks, 1e-6 if stem == "v2" else None))
        for i in (2, 3, 4):
            setattr(self, f"fuse{i}", M.SplitFusion(d))
        dpr = [v.item() for v in torch.linspace(0, config.drop_path, config.depth)]
        sizes = (2, 2, config.depth - 6, 2)          # [2,2,2,2] at depth 8, [2,2,6,2] at depth 12
        k = 0
        for si, n in enumerate(sizes, start=1):
            blocks = nn.ModuleList([
                M.QuadBlockWithTokenLearner(config, self.global_bank, dpr[k + j], config.use_token_learner, self._rt)
                for j in range(n)])
            setattr(self, f"stage{si}_blocks", blocks)
            k += n
        self.norm = nn.LayerNorm(d)
        self.head = nn.Linear(d, config.num_classes)
        self._pos_site = K.new_site()
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.apply(_init_weights)

    def forward(self, x):
        # one model pass: the bank copy a branch's write leaves stays valid for the next BLOCK's first branch too (nothing but
        # the branches writes the bank inside a forward), so only the first block takes its own snapshot
        rt = self._rt
        rt.in_model, rt.snap = True, None
        try:
            return self._forward_impl(x)
        finally:
            rt.in_model, rt.snap = False, None

    def _forward_impl(self, x):
        self._check(x)
        cdt = self._dtype(x)
        self.patch_embed.proj.compute_dtype = cdt
        # CNN lateral path: channel-last on the HIP kernels (no MIOpen convolution anywhere).  It does not meet the token
        # path before fuse2, so it runs on a second HIP stream beside patch-embed + stage 1 (autograd replays each node on
        # its forward stream, so backward overlaps the same way; under hipGraph capture the fork/join become graph edges).
        main = side = None
        if _LATERAL_STREAM and x.is_cuda:
            main = torch.cuda.current_stream(x.device)
            side = _side_stream(x.device)
            side.wait_stream(main)
        with torch.autocast("cuda", enabled=False):
            def lateral():
                with (torch.cuda.stream(side) if side is not None else contextlib.nullcontext()):
                    K.Stamps.mark("lat.begin")
                    feats, (fh, fw) = self.cnn_stem.forward_tokens(x, cdt)
                    R_ = {}
                    for i, f in zip((2, 3, 4), feats):
                        a = getattr(self, f"lmfa{i}").forward_tokens(F.stamp(f, f"lat.feat{i}"), fh, fw)
                        R_[i] = F.stamp(getattr(self, f"rrcv{i}")(a, self.H, self.W), f"lat.R{i}")
                        if side is not None:                 # the token path joins scale by scale: fuse2 needs R2 only
                            ready[i] = torch.cuda.Event()
                            ready[i].record(side)
                return R_

            def stage1():
                K.Stamps.mark("tok.begin")
                T_ = F.stamp(self.patch_embed(x, self.pos_embed), "tok.embed")
                T_ = F.dropout(T_, self.pos_drop.p, self._pos_site, self.training)
                if _EARLY_FLUSH and side is not None and torch.is_grad_enabled() and T_.requires_grad:
                    T_ = F.FlushMarkFn.apply(T_)            # backward: the token path ends here while the lateral path's tail still runs
                T_ = self._sync(T_, "stage1_blocks")
                for blk in self.stage1_blocks:
                    T_ = blk(T_)
                return F.stamp(T_, "tok.stage1")

            def lateral_steps(R_):
                feats, (fh, fw) = yield from self.cnn_stem.forward_tokens_steps(x, cdt)
                for i, f in zip((2, 3, 4), feats):
                    a = getattr(self, f"lmfa{i}").forward_tokens(f, fh, fw)
                    yield
                    R_[i] = getattr(self, f"rrcv{i}")(a, self.H, self.W)
                    ready[i] = torch.cuda.Event()
                    ready[i].record(side)
                    yield

            def stage1_steps(out):
                T_ = self.patch_embed(x, self.pos_embed)
                T_ = F.dropout(T_, self.pos_drop.p, self._pos_site, self.training)
                if _EARLY_FLUSH and side is not None and torch.is_grad_enabled() and T_.requires_grad:
                    T_ = F.FlushMarkFn.apply(T_)            # backward: the token path ends here while the lateral path's tail still runs
                T_ = self._sync(T_, "stage1_blocks")
                yield
                for blk in self.stage1_blocks:
                    if getattr(blk, "use_token_learner", False):
                        T_ = blk.token_learner(T_)
                        yield
                        T_ = blk.quad_block(T_)
                        yield
                        T_ = blk.token_upmix(T_)
                        yield
                    else:
                        T_ = blk(T_)
                        yield
                out["T"] = T_

            # Issue order of the two independent chains (QAVIT_LATERAL_ORDER: 0 lateral first, 1 token path first, 2 alternating).  It is
            # also the hipGraph's node creation order.  It does not matter for the free-running step: the wall-clock stamps of
            # tools/chain_stamps.py show both chains starting within 10 us of each other whatever the order; the 1.4 ms late start a
            # rocprofv3 kernel trace shows is the host's packet enqueue under tracing.
            ready = {}
            scales = None
            if _LATERAL_ORDER in (3, 4) and side is not None and hasattr(self.cnn_stem, "forward_tokens_scales"):
                # Scale by scale: the lateral work of scale i + 1 (stem stage, LMFAdapter, RRCV) is BUILT after the token path's stage i.
                # The side stream runs the same kernels in the same order as with order 0 -- what changes is the age of the autograd
                # nodes: the engine runs the youngest ready node first, so with the whole lateral chain built up front every stem node
                # was older than every block and the stem's backward (0.8 ms at B = 1024) queued up at the very end of the pass, where
                # since round 4 it, not the token path, ended the step (tools/chain_stamps.py: stem reached at 8.10 ms, patch embedding
                # at 7.91 ms).  Built per scale, stem stage i + 1's backward follows LMFAdapter i + 1's at once, beside the token path's
                # stage-i backward.
                gen = self.cnn_stem.forward_tokens_scales(x, cdt)
                R, fhw = {}, {}

                def lateral_scale(i):
                    with torch.cuda.stream(side):
                        if i == 2:
                            K.Stamps.mark("lat.begin")
                            f, fhw["hw"] = next(gen)
                        else:
                            f = next(gen)
                        fh, fw = fhw["hw"]
                        a = getattr(self, f"lmfa{i}").forward_tokens(F.stamp(f, f"lat.feat{i}"), fh, fw)
                        R[i] = F.stamp(getattr(self, f"rrcv{i}")(a, self.H, self.W), f"lat.R{i}")
                        ready[i] = torch.cuda.Event()
                        ready[i].record(side)
                scales = lateral_scale
                if _LATERAL_ORDER == 4:
                    # the token path's first stage is built BEFORE the lateral chain's first scale.  With the lateral scale first (order 3)
                    # the replayed graph started the token path only when the lateral chain had finished scale 3 (stamps: tok.begin at
                    # 0.95 ms instead of 0.05 ms, step 9.72 ms against 9.37): the hipGraph executor's schedule follows node creation
                    # order more than the round-3 probe suggested when a branch is extended after its sibling was built
                    T = stage1()
                    lateral_scale(2)
                else:
                    lateral_scale(2)
                    T = stage1()
            elif _LATERAL_ORDER == 2 and side is not None:
                R, box = {}, {}
                gl, gm = lateral_steps(R), stage1_steps(box)
                live_l = live_m = True
                while live_l or live_m:
                    if live_l:
                        with torch.cuda.stream(side):
                            live_l = next(gl, _DONE) is not _DONE
                    if live_m:
                        live_m = next(gm, _DONE) is not _DONE
                T = box["T"]
            elif _LATERAL_ORDER == 1 and side is not None:
                T = stage1()
                R = lateral()
            else:
                R = lateral()
                T = stage1()
            for si in (2, 3, 4):
                if side is not None:
                    # R2 is complete ~0.5 ms before R4 (tools/chain_stamps.py): waiting for the whole lateral chain here left the
                    # token path idle for 0.45 ms per forward.  The last event is the last work on the side stream, so the chains
                    # are fully joined (and a capture is closed) once fuse4 has waited.
                    main.wait_event(ready[si])
                    R[si].record_stream(main)
                T = self._sync(T, f"fuse{si}")
                T = F.stamp(getattr(self, f"fuse{si}")(T, R[si]), f"tok.fuse{si}")
                T = self._sync(T, f"stage{si}_blocks")
                for blk in getattr(self, f"stage{si}_blocks"):
                    T = blk(T)
                if scales is not None and si < 4:
                    scales(si + 1)
            return self._head(T)


class QAViT(_Base):
    """variant 'v1' = QAViT.py, 'v2' = QAViTv2.py (stabilised CCF-FFN + depthwise bias + bank update_count),
    'hqa' = QAViTv2_CIFAR100.py (v2 without the depthwise bias)."""

    def __init__(self, config: QAViTConfig, variant: str = "v1"):
        super().__init__()
        self.config = config
        self._rt = M._Ctx(variant)
        self.num_patches = (config.img_size // config.patch_size) ** 2
        d = config.embed_dim
        self.patch_embed = M.PatchEmbed(config.img_size, config.patch_size, config.in_channels, d)
        self.pos_embed = nn.Parameter(torch.zeros(1, self.num_patches, d))
        self.pos_drop = nn.Dropout(config.dropout)
        self.global_bank = M.GlobalTokenBank(config.global_bank_size, d, with_counter=(variant != "v1"))
        dpr = [v.item() for v in torch.linspace(0, config.drop_path, config.depth)]
        self.blocks = nn.ModuleList([M.QuadAttentionBlock(config, self.global_bank, dpr[i], self._rt) for i in range(config.depth)])
        self.norm = nn.LayerNorm(d)
        self.head = nn.Linear(d, config.num_classes)
        self._pos_site = K.new_site()
        nn.init.trunc_normal_(self.pos_embed, std=0.02)
        self.apply(_init_weights)

    def forward(self, x):
        # one model pass: the bank copy a branch's write leaves stays valid for the next BLOCK's first branch too (nothing but
        # the branches writes the bank inside a forward), so only the first block takes its own snapshot
        rt = self._rt
        rt.in_model, rt.snap = True, None
        try:
            return self._forward_impl(x)
        finally:
            rt.in_model, rt.snap = False, None

    def _forward_impl(self, x):
        self._check(x)
        cdt = self._dtype(x)
        self.patch_embed.proj.compute_dtype = cdt
        with torch.autocast("cuda", enabled=False):
            T = self.patch_embed(x, self.pos_embed)
            T = F.dropout(T, self.pos_drop.p, self._pos_site, self.training)
            for i, blk in enumerate(self.blocks):
                if i > 0 and i % 2 == 0:                   # data-parallel: gradients of blocks >= i are complete once backward passes here
                    T = self._sync(T, f"blocks.{i}")
                T = blk(T)
            return self._head(T)
