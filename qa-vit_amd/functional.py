"""Autograd operators of the QA-ViT hot path, each a thin torch.autograd.Function over the HIP kernels.

Conventions
  * activations are fp32 or bf16 (the "compute dtype"); parameters stay fp32 (master weights); 2-D weights
    are read through a WeightPack (compute-dtype copy + transposed copy, refreshed by one kernel launch);
  * parameter gradients are accumulated by the kernels DIRECTLY into ``param.grad`` (fp32, created on
    demand) and the Function returns None for them -- no per-parameter add kernels.  ``.grad`` is therefore
    correct after ``loss.backward()``, which is what the reference's training loops consume;
  * dropout / drop-path masks are regenerated in backward from (seed, step, site, index).
"""
import ctypes as C
import math
import os
import weakref
from typing import List, Optional

import torch
from torch.autograd import Function

from . import kernels as K
from . import lib as L


# ---------------------------------------------------------------------------------------------------
# packed weights
# ---------------------------------------------------------------------------------------------------
class _PackEntry:
    __slots__ = ("refs", "versions", "ptrs", "dst", "dstT", "rows", "cols", "frag")

    def params(self):
        """The live parameter tensors, or None once any of them has been garbage-collected."""
        ps = [r() for r in self.refs]
        return None if any(p is None for p in ps) else ps


class WeightPack:
    """Compute-dtype copies of 2-D weights: ``W`` (as stored, [N,K]) and ``W^T`` ([K,N]).

    One ``refresh()`` launch re-packs every registered weight (the descriptor table lives on the device),
    so a training step pays one kernel for all ~200 weights; it is also what a captured hipGraph replays
    at the top of each step.

    Staleness: a copy is stale when (i) a torch optimizer stepped (``_version`` moved), (ii) something wrote the
    parameters through raw pointers -- the fused AdamW kernel, a graph replay -- and called ``mark_stale()``, or
    (iii) the parameter's storage was re-bound (``p.data = ...``, ``.to()``; the descriptor table caches ``data_ptr``).
    ``get()`` checks all three, so ``model.eval()(x)`` right after ``Trainer.step()`` / ``replay()`` sees the new weights.
    Entries hold weak references: a dead model's weights are dropped at the next refresh."""

    def __init__(self, device):
        self.device = device
        self.entries = {}      # (ids, dtype) -> _PackEntry
        self._tables = {}      # dtype -> (device bytes tensor, n_desc, max_elems)
        self._retired = []     # descriptor tables a captured hipGraph may still replay: kept alive, never reused
        self._stale = False

    @staticmethod
    def _as2d(w):
        return w.reshape(w.shape[0], -1)

    def mark_stale(self):
        """The parameters were written behind autograd's back (raw-pointer optimiser kernel, graph replay)."""
        self._stale = True

    def _make(self, params, dtype, frag=0):
        e = _PackEntry()
        e.refs = [weakref.ref(p) for p in params]
        rows = sum(p.shape[0] for p in params)
        cols = self._as2d(params[0]).shape[1]
        e.rows, e.cols = rows, cols
        e.frag = frag
        if frag:                                           # MFMA fragment order (qavit_pack_desc.pad = 1): one buffer, no transpose
            if any(p.shape[0] % 16 for p in params) or cols % 32:
                raise RuntimeError("fragment-packed weights need rows % 16 == 0 and cols % 32 == 0")
            e.dst = torch.empty(rows * cols, dtype=dtype, device=self.device)
            e.dstT = None
            e.versions = [-1] * len(params)
            e.ptrs = [p.data_ptr() for p in params]
            return e
        single_f32 = dtype == torch.float32 and len(params) == 1
        e.dst = None if single_f32 else torch.empty(rows, cols, dtype=dtype, device=self.device)
        e.dstT = torch.empty(cols, rows, dtype=dtype, device=self.device)
        e.versions = [-1] * len(params)
        e.ptrs = [p.data_ptr() for p in params]
        return e

    def _descs(self, e, params):
        out, off = [], 0
        esz = (e.dstT if e.dstT is not None else e.dst).element_size()
        for p in params:
            r = p.shape[0]
            d = L.PackDesc()
            d.src = self._as2d(p).data_ptr()
            d.dst = 0 if e.dst is None else e.dst.data_ptr() + off * e.cols * esz
            d.dstT = 0 if e.dstT is None else e.dstT.data_ptr() + off * esz
            d.rows, d.cols, d.ldT, d.pad = r, e.cols, e.rows, int(e.frag)
            out.append(d)
            off += r
        e.ptrs = [p.data_ptr() for p in params]
        return out

    def _launch(self, descs, dtype):
        arr = (L.PackDesc * len(descs))(*descs)
        raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
        mx = max(d.rows * d.cols for d in descs)
        K.pack_weights(dtype, raw, len(descs), mx)
        return raw, len(descs), mx

    def _drop_table(self, dtype):
        tab = self._tables.pop(dtype, None)
        if tab is not None:
            self._retired.append(tab[0])

    def refresh(self, dtype=None):
        """Re-pack every registered weight (one launch per compute dtype in use)."""
        dts = {k[1] for k in self.entries} if dtype is None else {dtype}
        for dt in dts:
            live = {}
            moved = False
            for key, e in list(self.entries.items()):
                if key[1] != dt:
                    continue
                ps = e.params()
                if ps is None:                                  # the model was freed
                    del self.entries[key]
                    moved = True
                    continue
                live[key] = ps
                moved |= any(a != p.data_ptr() for a, p in zip(e.ptrs, ps))
            if moved:
                self._drop_table(dt)
            tab = self._tables.get(dt)
            if tab is None:
                descs = []
                for key, ps in live.items():
                    descs.extend(self._descs(self.entries[key], ps))
                if not descs:
                    continue
                self._tables[dt] = self._launch(descs, dt)
            else:
                K.pack_weights(dt, tab[0], tab[1], tab[2])
            for key, ps in live.items():
                self.entries[key].versions = [p._version for p in ps]
        if dtype is None:
            self._stale = False

    def get(self, params, dtype):
        """-> (W [rows, cols], W^T [cols, rows]) in ``dtype`` for a weight or a row-stack of weights."""
        if isinstance(params, torch.Tensor):
            params = [params]
        if self._stale:
            self.refresh()
        key = (tuple(id(p) for p in params), dtype)
        e = self.entries.get(key)
        if e is not None and e.params() is None:               # an id re-used by a new tensor after the old one died
            del self.entries[key]
            self._drop_table(dtype)
            e = None
        if e is None:
            e = self._make(params, dtype)
            self.entries[key] = e
            self._drop_table(dtype)
            self._launch(self._descs(e, params), dtype)          # first use: pack just this one
            e.versions = [p._version for p in params]
        elif any(v != p._version for v, p in zip(e.versions, params)) or any(a != p.data_ptr() for a, p in zip(e.ptrs, params)):
            self.refresh(dtype)                           # an optimizer stepped / storage was re-bound: re-pack everything once
        W = e.dst if e.dst is not None else self._as2d(params[0]).detach()
        return W, e.dstT

    def get_frag(self, params, dtype=torch.bfloat16, order=1):
        """-> the weight (or row-stack of weights) in MFMA fragment order (include/qavit.h, qavit_pack_desc.pad = ``order``:
        1 = plain k order, 2 = paired-quads k order): the operand image of the fused branch kernels.  Refreshed with
        everything else."""
        if isinstance(params, torch.Tensor):
            params = [params]
        if self._stale:
            self.refresh()
        key = (tuple(id(p) for p in params), dtype, "frag%d" % order)
        e = self.entries.get(key)
        if e is not None and e.params() is None:
            del self.entries[key]
            self._drop_table(dtype)
            e = None
        if e is None:
            e = self._make(params, dtype, frag=order)
            self.entries[key] = e
            self._drop_table(dtype)
            self._launch(self._descs(e, params), dtype)
            e.versions = [p._version for p in params]
        elif any(v != p._version for v, p in zip(e.versions, params)) or any(a != p.data_ptr() for a, p in zip(e.ptrs, params)):
            self.refresh(dtype)
        return e.dst


_packs = {}


def pack_for(device) -> WeightPack:
    key = torch.device(device).index or 0
    if key not in _packs:
        _packs[key] = WeightPack(torch.device("cuda", key))
    return _packs[key]


# ---------------------------------------------------------------------------------------------------
# gradient sinks
# ---------------------------------------------------------------------------------------------------
def grad_sink(t: Optional[torch.Tensor]):
    """-> (fp32 accumulation buffer, value to return from backward).  Leaf parameters accumulate in
    place into ``.grad``; anything else gets a fresh zero buffer that is returned to autograd."""
    if t is None or not t.requires_grad:
        return None, None
    if t.is_leaf:
        if t.grad is None:
            t.grad = torch.zeros_like(t, dtype=torch.float32)
        return t.grad, None
    buf = torch.zeros(t.shape, dtype=torch.float32, device=t.device)
    return buf, buf


def _ret(buf_ret, like):
    if buf_ret is None:
        return None
    return buf_ret if buf_ret.dtype == like.dtype else buf_ret.to(like.dtype)


def _rt(x):
    return K.Runtime.get(x.device)


class DeferDW:
    """Defer the weight-gradient GEMMs of a backward pass and launch them in grouped kernels (kernels.DeferredTN).
    The first deferral of a pass registers an autograd-engine callback that flushes when backward ends, so
    ``loss.backward(); optimizer.step()`` sees complete gradients; ``flush()`` can be called earlier (DDP sync points)."""
    enabled = os.environ.get("QAVIT_DEFER_DW", "1") != "0"
    _armed = False

    @classmethod
    def arm(cls):
        if not cls.enabled or cls._armed:
            return cls.enabled
        try:
            torch.autograd.Variable._execution_engine.queue_callback(cls.finish)
        except RuntimeError:
            return False                                    # not inside a backward pass
        cls._armed = True
        K.DeferredTN.enabled = True
        K.DeferredLN.enabled = True
        K.DeferredTN.home_stream = K.stream()                # backward starts on the caller's stream; only its dW GEMMs are queued
        return True

    _side = {}

    @classmethod
    def _launch(cls, rng=None):
        """Launch what is queued.  ``rng`` = (lo, hi): a data-parallel sync point -- only the weight-gradient GEMMs whose destination lies in
        that address range of the flat gradient buffer (the bucket prefix about to be all-reduced) and this stream's partial rows; everything
        else waits for the end of the pass."""
        home = rng is not None

        def tn_flush():
            if home:
                K.DeferredTN.flush_range(*rng)
            else:
                K.DeferredTN.flush()

        def small():
            if DeferredBank.queue:
                K.DeferredLN.flush(home_only=home)           # the projected bank rows' gradients are among the partial rows
                DeferredBank.run()                           # queues the projections' own (fp32, tiny) weight gradients
                tn_flush()
            K.DeferredLN.flush(home_only=home)

        # (sync points keep the one-stream order: measured, seven forks / joins per backward cost more than the overlap returns)
        if home or not (_FLUSH_SIDE and torch.cuda.is_available() and K.DeferredTN.queue and not K.DeferredTN.ASYNC):
            if DeferredBank.queue:
                K.DeferredLN.flush(home_only=home)
                DeferredBank.run()
            tn_flush()
            K.DeferredLN.flush(home_only=home)
            K.DeferredTN.join()
            return
        # ~30 small launches (five partial-row reduces, the bank projections' sixteen skinny GEMMs and their fp32 weight gradients) used to
        # run one after the other in front of the one-launch weight-gradient kernel -- 0.25 ms of a tail in which nothing else runs.  None of
        # them touches what that kernel reads or writes: they go to a second stream forked HERE (before the big launch is queued) and joined
        # after it, so they run beside its first workgroups instead of in front of them.
        dev = torch.cuda.current_device()
        main = torch.cuda.current_stream(dev)
        side = cls._side.get(dev)
        if side is None:
            side = cls._side[dev] = torch.cuda.Stream(device=dev)
        fork = torch.cuda.Event()
        fork.record(main)
        tn_flush()                                           # this stream: the weight-gradient GEMMs queued during the pass (in range)
        with torch.cuda.stream(side):
            side.wait_event(fork)
            home_was = K.DeferredTN.home_stream
            K.DeferredTN.home_stream = K.stream()            # what the bank projections queue here belongs to THIS pass's own stream, not to a foreign chain
            try:
                small()
            finally:
                K.DeferredTN.home_stream = home_was
        main.wait_stream(side)
        K.DeferredTN.join()

    @classmethod
    def flush(cls):
        """Launch what is queued and wait for it (data-parallel sync points: the bucket about to be reduced must be complete)."""
        cls._launch()

    @classmethod
    def flush_range(cls, lo: int, hi: int):
        """Data-parallel sync point: complete the gradients stored in [lo, hi) of the flat gradient buffer (the bucket prefix about to be
        all-reduced) and nothing else.  The bank projections' deferred backward and this stream's LayerNorm partial rows are cheap single
        launches and are run whole; weight-gradient GEMMs outside the range and everything another stream queued wait for the end."""
        if not cls._armed:
            return
        cls._launch((lo, hi))

    @classmethod
    def flush_home(cls):
        """Launch what the arming stream has queued so far and leave the other streams' entries for the end of the pass: called where
        this stream's backward is (almost) over while another stream's chain still runs (HQAViT: the CNN lateral path's backward
        tail), so the weight-gradient GEMMs run beside that chain instead of after it."""
        if not cls._armed:
            return
        if DeferredBank.queue:
            K.DeferredLN.flush(home_only=True)
            DeferredBank.run()
        K.DeferredTN.flush(home_only=True)
        K.DeferredLN.flush(home_only=True)

    @classmethod
    def finish(cls):
        K.Stamps.mark("dw.flush")
        cls._launch()
        K.DeferredTN.enabled = False
        K.DeferredLN.enabled = False
        cls._armed = False


class FlushMarkFn(Function):
    """Identity whose backward launches the deferred weight-gradient work queued so far on this stream (DeferDW.flush_home)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        DeferDW.flush_home()
        return dy


class StampFn(Function):
    """Identity that takes a wall-clock stamp ``name + ".f"`` when forward reaches it and ``name + ".b"`` when backward does (K.Stamps)."""

    @staticmethod
    def forward(ctx, x, name):
        ctx.name = name
        K.Stamps.mark(name + ".f")
        return x.view_as(x)

    @staticmethod
    def backward(ctx, dy):
        K.Stamps.mark(ctx.name + ".b")
        return dy, None


def stamp(x, name):
    return StampFn.apply(x, name) if K.Stamps.enabled else x


class SideStream:
    """Weight-gradient GEMMs are independent of the activation-gradient chain: they run on a second HIP stream
    (fork after their inputs exist, join before anyone reads ``.grad``) so their latency hides under the dX /
    LayerNorm-backward kernels.  Works eagerly and under hipGraph capture (fork/join become graph edges)."""
    enabled = os.environ.get("QAVIT_DW_STREAM", "0") != "0"   # measured slower under hipGraph (fork/join edges): opt-in
    _streams = {}
    _dirty = {}
    _main = {}

    @classmethod
    def get(cls, device):
        key = torch.device(device).index or 0
        st = cls._streams.get(key)
        if st is None:
            st = torch.cuda.Stream(device=device)
            cls._streams[key] = st
        return st

    @classmethod
    def fork(cls, device, *tensors):
        """-> context manager running on the side stream after everything enqueued so far on the current stream."""
        st = cls.get(device)
        key = torch.device(device).index or 0
        main = torch.cuda.current_stream(device)
        st.wait_stream(main)
        for t in tensors:
            if t is not None:
                t.record_stream(st)
        if not cls._dirty.get(key):
            # first fork of this backward pass: join automatically when the autograd engine finishes, on the
            # stream the backward kernels run on, so `loss.backward(); optimizer.step()` stays correct
            cls._dirty[key] = True
            cls._main[key] = main
            try:
                torch.autograd.Variable._execution_engine.queue_callback(lambda d=device: cls.join(d))
            except RuntimeError:
                pass                                        # not inside a backward pass: caller joins explicitly
        return torch.cuda.stream(st)

    @classmethod
    def join(cls, device):
        """Make the current stream wait for all side-stream work (call before gradients are consumed)."""
        key = torch.device(device).index or 0
        if cls._dirty.get(key):
            main = cls._main.get(key) or torch.cuda.current_stream(device)
            main.wait_stream(cls.get(device))
            cur = torch.cuda.current_stream(device)
            if cur != main:
                cur.wait_stream(cls.get(device))
            cls._dirty[key] = False


# ---------------------------------------------------------------------------------------------------
# Linear (+ fused LayerNorm prologue, GELU / dropout / drop-path / residual epilogue)
# ---------------------------------------------------------------------------------------------------
_FLUSH_SIDE = os.environ.get("QAVIT_FLUSH_SIDE", "1") != "0"    # end of backward: the small reduce / bank launches beside the one-launch weight-gradient kernel
DEFER_FIX_CFUSE = os.environ.get("QAVIT_DEFER_NANFIX_CFUSE", "1") != "0"   # the cross branch's NaN rule rides in the compress-fuse launch
_DEFER_FIX = os.environ.get("QAVIT_DEFER_NANFIX", "1") != "0"   # fused branches followed by a bank write: the NaN rule's rewrite rides in the bank-statistics launch
_UPMIX_FWD_SA = os.environ.get("QAVIT_UPMIX_FWD_SA", "1") != "0"   # block tail + up-mix forward: the scale-add formed while the up-mix stages the image
MIX3_LN = os.environ.get("QAVIT_MIX3_LN", "1") != "0"     # SplitFusion: blend + final LayerNorm as one launch each way
GATE_MIX3_LN = os.environ.get("QAVIT_GATE_MIX3_LN", "1") != "0"   # ... and the gate in front of the blend with them
_LINEAR_CAT = os.environ.get("QAVIT_LINEAR_CAT", "1") != "0"   # Linear on cat([T, R]): one two-source GEMM instead of a GEMM + an accumulating GEMM
_LN_LIN = os.environ.get("QAVIT_LN_LIN", "1") != "0"      # narrow LayerNorm-prologue Linears: dX GEMM fused into the LayerNorm-backward kernel


class LinearFn(Function):
    """y = resid + droppath(dropout(act(LN(x) @ W[rows]^T + b[rows])))"""

    @staticmethod
    def forward(ctx, x, w, b, ln_g, ln_b, resid, opts):
        K._require_cuda(x, w)
        rt = _rt(x)
        Kd = x.shape[-1]
        x2 = x.reshape(-1, Kd)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        off, n = opts.get("rows") or (0, w.shape[0])
        Kfull = w.shape[1] if w.dim() == 2 else w[0].numel()
        koff = (opts.get("cols") or (0, Kfull))[0]          # cols = (k_off, k_len): y = x @ w[rows, k_off:k_off+k_len]^T (x has k_len columns)
        lda = Kd
        if opts.get("xpad"):                                # x's rows carry zero columns beyond the weight's K (Im2ColFn): no gradient flows back
            if ln_g is not None or opts.get("cols") or x.requires_grad or Kd < Kfull:
                raise ValueError("xpad: plain Linear on a non-differentiable, zero-padded input only")
            Kd = Kfull
        Wc, Wt = pack_for(x.device).get(w, x.dtype)
        y = torch.empty(M, n, dtype=x.dtype, device=x.device)
        act = 1 if opts.get("act") == "gelu" else 0
        drop = opts.get("drop") or (0.0, 0)
        dp = opts.get("dp") or (0.0, 0, 1)
        Z = torch.empty_like(y) if (act and any(ctx.needs_input_grad)) else None
        ln = None
        stats = None
        if ln_g is not None:
            ln = (ln_g, ln_b, opts.get("eps", 1e-5))
            stats = (torch.empty(M, dtype=torch.float32, device=x.device), torch.empty(M, dtype=torch.float32, device=x.device))
            # a_mode 3: the GEMM computes the row statistics in its own prologue (or launches row_stats itself)
        r2 = None
        if resid is not None:
            r2 = resid.reshape(-1, n)
            if not r2.is_contiguous():
                r2 = r2.contiguous()
        esz = Wc.element_size()
        bias_ptr = None if b is None else b.data_ptr() + off * 4
        a = dict(a_mode=3 if ln else 0, ln=ln, ln_stats=stats, Z=Z, act=act, drop=drop, dp=dp, R=r2, ldr=n, rng=rt.rng)
        # bias pointer offset: pass a narrow view tensor to keep kernels.gemm_nt simple
        bview = None if b is None else b.detach()[off:off + n]
        K.gemm_nt(x2, Wc, y, M, n, Kd, lda, Kfull, n, bview, B_ptr=Wc.data_ptr() + (off * Kfull + koff) * esz, **a)
        ctx.opts = dict(opts)
        ctx.meta = (M, n, Kd, off, act, drop, dp, x.shape, resid is not None, koff)
        ctx.save_for_backward(x2, w, b, ln_g, ln_b, Z, stats[0] if stats else None, stats[1] if stats else None)
        if opts.get("alias"):
            ctx.set_materialize_grads(False)
            return y.reshape(*x.shape[:-1], n), x.view_as(x)
        return y.reshape(*x.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy, dalias=None):
        x2, w, b, ln_g, ln_b, Z, mean, rstd = ctx.saved_tensors
        M, n, Kd, off, act, drop, dp, xshape, has_res, koff = ctx.meta
        if dy is None:                                      # only the alias was differentiated through
            return dalias, None, None, None, None, None, None
        dx_add = None
        if dalias is not None and ln_g is None and dalias.dtype == x2.dtype:
            dx_add = dalias.reshape(M, Kd)
            if not dx_add.is_contiguous():
                dx_add = dx_add.contiguous()
            dalias = None
        dx = _linear_bwd(x2, w, b, dy.reshape(M, n), off, n, ctx.needs_input_grad[0], ln_g, ln_b, Z, mean, rstd, act, drop, dp, koff,
                         dres=dalias, dx_add=dx_add, kd=Kd)
        dres = dy if has_res else None
        return (dx.reshape(xshape) if (dx is not None and ctx.needs_input_grad[0]) else None), None, None, None, None, dres, None


class LinearCatFn(Function):
    """y = cat([x1, x2], -1) @ W^T + b without the cat: ONE GEMM whose A operand switches source at column K1 (qavit_gemm_args.A2) --
    SplitFusion's Linear(2C -> C) on cat([T, R]) (HQAViT_CIFAR100.py:951), which was one GEMM on T and a second, accumulating one on R
    (its output written, read back and written again).  Backward = that of the two Linears (two input-gradient GEMMs on W's column
    halves, two deferred weight-gradient problems, one bias column sum)."""

    @staticmethod
    def forward(ctx, x1, x2, w, b):
        K._require_cuda(x1, w)
        K1, K2 = x1.shape[-1], x2.shape[-1]
        a1 = x1.reshape(-1, K1)
        a2 = x2.reshape(-1, K2)
        if not a1.is_contiguous():
            a1 = a1.contiguous()
        if not a2.is_contiguous():
            a2 = a2.contiguous()
        M, n = a1.shape[0], w.shape[0]
        Wc, _ = pack_for(x1.device).get(w, x1.dtype)
        y = torch.empty(M, n, dtype=x1.dtype, device=x1.device)
        K.gemm_nt(a1, Wc, y, M, n, K1 + K2, K1, K1 + K2, n, None if b is None else b.detach(), A2=a2, lda2=K2, a2_k0=K1)
        ctx.save_for_backward(a1, a2, w, b)
        ctx.shapes = (x1.shape, x2.shape)
        return y.reshape(*x1.shape[:-1], n)

    @staticmethod
    def backward(ctx, dy):
        a1, a2, w, b = ctx.saved_tensors
        M, n = a1.shape[0], w.shape[0]
        dy2 = dy.reshape(M, n)
        d1 = _linear_bwd(a1, w, b, dy2, 0, n, ctx.needs_input_grad[0], koff=0)
        d2 = _linear_bwd(a2, w, None, dy2, 0, n, ctx.needs_input_grad[1], koff=a1.shape[1])
        return (d1.reshape(ctx.shapes[0]) if (d1 is not None and ctx.needs_input_grad[0]) else None,
                d2.reshape(ctx.shapes[1]) if (d2 is not None and ctx.needs_input_grad[1]) else None, None, None)


def linear_cat_ok(x1, x2, w) -> bool:
    """Can LinearCatFn take cat([x1, x2]) @ w^T?  (bf16 operands of the same row count, a split on a 64-column boundary, a shape of the K-loop GEMM)"""
    if not (_LINEAR_CAT and x1.is_cuda and x1.dtype == x2.dtype and x1.shape[:-1] == x2.shape[:-1] and w.dim() == 2
            and w.shape[1] == x1.shape[-1] + x2.shape[-1]):
        return False
    M = x1.numel() // x1.shape[-1]
    return K.gemm_nt_a2_ok(x1, x2, M, w.shape[0], w.shape[1], x1.shape[-1]) and x1.shape[-1] % 8 == 0 and x2.shape[-1] % 8 == 0


def _linear_bwd(x2, w, b, dy2, off, n, need_dx, ln_g=None, ln_b=None, Z=None, mean=None, rstd=None, act=0, drop=(0.0, 0), dp=(0.0, 0, 1), koff=0,
                dres=None, dx_add=None, kd=None):
    """Backward of y = droppath(dropout(act(LN(x2) @ w[off:off+n]^T + b[off:off+n]))) for row matrices: returns dx (or None) and
    accumulates dW / db into the parameters' .grad (deferred grouped weight-gradient GEMMs).  Shared by LinearFn and BranchFn.
    ``kd``: the contraction length when x2's rows are zero-padded beyond it (LinearFn xpad; no dx)."""
    M, ldx = x2.shape
    Kd = kd or ldx
    if Kd != ldx and (need_dx or ln_g is not None):
        raise ValueError("padded input rows: weight / bias gradients only")
    rt = _rt(x2)
    if not dy2.is_contiguous():
        dy2 = dy2.contiguous()
    need_t = bool(act) or drop[0] > 0.0 or dp[0] > 0.0
    need_dw = w.requires_grad
    Wc, Wt = pack_for(x2.device).get(w, x2.dtype)
    esz = Wt.element_size()
    dz = dy2
    dx = None
    if (_LN_LIN and need_dx and not need_t and ln_g is not None and dx_add is None and koff == 0 and Kd == ldx
            and K.layernorm_bwd_lin_ok(x2, dy2, n, Kd) and (dres is None or K.ln_dres_ok(x2, dres.reshape(M, Kd), Kd))):
        # narrow Linear behind a LayerNorm (TokenLearner's 192 -> 16 scores): its input-gradient GEMM redone inside the LayerNorm-backward
        # kernel -- no [M, Kd] product written by one launch and read back by the next
        gbuf, _ = grad_sink(ln_g)
        bbuf, _ = grad_sink(ln_b)
        dx = torch.empty(M, Kd, dtype=x2.dtype, device=x2.device)
        K.layernorm_bwd_lin(dy2, Wc.data_ptr() + off * Wc.shape[1] * Wc.element_size(), Wc.shape[1], n, x2, ln_g, mean, rstd, dx, gbuf, bbuf, M, Kd,
                            dres=None if dres is None else dres.reshape(M, Kd))
    elif need_dx or need_t:
        dz = torch.empty_like(dy2) if (need_t and (need_dw or (b is not None and b.requires_grad))) else dy2
        bwd = None
        if need_t:
            bwd = dict(Z=Z, ldz=n, act=act, drop=drop, dp=dp, out=dz if dz is not dy2 else None, ldo=n)
        r2 = None if dres is None else dres.reshape(M, Kd)
        if (_LN_EPI and ln_g is not None and need_dx and dx_add is None and koff == 0 and Kd == ldx
                and K.gemm_nt_lnbwd_ok(x2, M, Kd, n, 2 if need_t else 0, r2) and dy2.data_ptr() % 16 == 0):
            # the LayerNorm backward in the input-gradient GEMM's epilogue (the K-loop kernel's column block is the whole LayerNorm row):
            # no [M, Kd] product written by one launch and read back by the next
            gbuf, _ = grad_sink(ln_g)
            bbuf, _ = grad_sink(ln_b)
            dx = torch.empty(M, Kd, dtype=x2.dtype, device=x2.device)
            DeferDW.arm()
            K.gemm_nt(dy2, Wt, dx, M, Kd, n, n, Wt.shape[1], Kd, None, a_mode=2 if need_t else 0, bwd=bwd, rng=rt.rng,
                      B_ptr=Wt.data_ptr() + (koff * Wt.shape[1] + off) * esz, R=r2, ldr=Kd,
                      lnbwd=dict(x=x2, mean=mean, rstd=rstd, gamma=ln_g, dgamma=gbuf, dbeta=bbuf))
            return _linear_bwd_weights(x2, w, b, dz, off, n, Kd, ldx, koff, ln_g, ln_b, mean, rstd, need_dw, dx)
        dxn = torch.empty(M, Kd, dtype=x2.dtype, device=x2.device)
        # dx_add [M, Kd]: another gradient of the same x, added in this GEMM's residual epilogue (no LayerNorm behind it)
        K.gemm_nt(dy2, Wt, dxn, M, Kd, n, n, Wt.shape[1], Kd, None, a_mode=2 if need_t else 0, bwd=bwd, rng=rt.rng,
                  B_ptr=Wt.data_ptr() + (koff * Wt.shape[1] + off) * esz, R=dx_add if ln_g is None else None, ldr=Kd)
        if ln_g is not None:
            gbuf, _ = grad_sink(ln_g)
            bbuf, _ = grad_sink(ln_b)
            dx = torch.empty_like(dxn)
            r2 = None
            if dres is not None:
                r2 = dres.reshape(M, Kd)
                if not K.ln_dres_ok(x2, r2, Kd):
                    r2 = None
            K.layernorm_bwd(dxn, x2, ln_g, mean, rstd, dx, gbuf, bbuf, M, Kd, dres=r2)
            if dres is not None and r2 is None:
                dx = dx + dres.reshape(M, Kd)
        else:
            dx = dxn if dres is None else dxn + dres.reshape(M, Kd)
    return _linear_bwd_weights(x2, w, b, dz, off, n, Kd, ldx, koff, ln_g, ln_b, mean, rstd, need_dw, dx)


_LN_EPI = os.environ.get("QAVIT_LN_EPILOGUE", "1") != "0"   # LayerNorm backward as the input-gradient GEMM's epilogue (csrc/gemm_big.hip, EPI 2)


def _linear_bwd_weights(x2, w, b, dz, off, n, Kd, ldx, koff, ln_g, ln_b, mean, rstd, need_dw, dx):
    """The weight / bias gradient half of _linear_bwd (deferred grouped GEMMs); returns ``dx`` unchanged."""
    M = x2.shape[0]
    if need_dw or (b is not None and b.requires_grad):
        wbuf, _ = grad_sink(w)
        bbuf2, _ = grad_sink(b)
        if wbuf is None:   # bias-only gradient: still use the kernel with a scratch C
            wbuf = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
        lnarg = (ln_g, ln_b, mean, rstd) if ln_g is not None else None
        Kfull = w.numel() // w.shape[0]
        DeferDW.arm()
        if SideStream.enabled:
            with SideStream.fork(x2.device, dz, x2, mean, rstd):
                K.gemm_tn(dz, x2, wbuf, M, n, Kd, n, ldx, Kfull, None, ln=lnarg,
                          C_ptr=wbuf.data_ptr() + (off * Kfull + koff) * 4,
                          colsum_ptr=None if bbuf2 is None else bbuf2.data_ptr() + off * 4)
        else:
            K.gemm_tn(dz, x2, wbuf, M, n, Kd, n, ldx, Kfull, None, ln=lnarg,
                      C_ptr=wbuf.data_ptr() + (off * Kfull + koff) * 4,
                      colsum_ptr=None if bbuf2 is None else bbuf2.data_ptr() + off * 4)
    return dx


def linear(x, w, b=None, *, ln=None, act=None, drop=None, dp=None, resid=None, rows=None, cols=None, eps=1e-5, train=True, alias=False, xpad=False):
    """``rows=(off, n)``: output rows off..off+n of ``w``; ``cols=(k_off, k_len)``: the K-slice w[:, k_off:k_off+k_len] (x has k_len
    columns) -- with ``resid`` it turns a Linear on a concatenation into accumulating GEMMs, no ``cat`` buffer."""
    ln_g, ln_b = ln if ln is not None else (None, None)
    # alias=True: -> (y, x_alias); the gradient that arrives on x_alias (x's other consumer) is added inside the LayerNorm-backward
    # kernel (LayerNorm-prologue Linears) or by the input-gradient GEMM's residual epilogue, instead of by an elementwise add of autograd's
    want_alias = bool(alias) and torch.is_grad_enabled() and x.requires_grad
    opts = dict(act=act, drop=drop, dp=dp, rows=rows, cols=cols, eps=eps, train=train, alias=want_alias, xpad=xpad)
    out = LinearFn.apply(x, w, b, ln_g, ln_b, resid, opts)
    if alias and not want_alias:
        return out, x
    return out


class LinearStack3Fn(Function):
    """y = x @ [W1;W2;W3]^T + [b1;b2;b3] (CGA q/k/v projections, HQAViT_CIFAR100.py:566-568) as ONE GEMM."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, w3, b3):
        Kd = x.shape[-1]
        x2 = x.reshape(-1, Kd)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        M = x2.shape[0]
        ws = [w1, w2, w3]
        Wc, Wt = pack_for(x.device).get(ws, x.dtype)
        n = Wc.shape[0]
        bias = pack_for(x.device).get([b1, b2, b3], torch.float32)[0].reshape(-1)       # stacked by the per-step pack launch, not a cat per call
        y = torch.empty(M, n, dtype=x.dtype, device=x.device)
        K.gemm_nt(x2, Wc, y, M, n, Kd, Kd, Kd, n, bias)
        ctx.save_for_backward(x2, w1, b1, w2, b2, w3, b3)
        ctx.xshape = x.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x2, w1, b1, w2, b2, w3, b3 = ctx.saved_tensors
        M, Kd = x2.shape
        ws, bs = [w1, w2, w3], [b1, b2, b3]
        Wc, Wt = pack_for(x2.device).get(ws, x2.dtype)
        n = Wc.shape[0]
        dy2 = dy.reshape(M, n)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, Kd, dtype=x2.dtype, device=x2.device)
            K.gemm_nt(dy2, Wt, dx, M, Kd, n, n, n, Kd, None)
        off = 0
        esz = dy2.element_size()
        for w, b in zip(ws, bs):
            ni = w.shape[0]
            wbuf, _ = grad_sink(w)
            bbuf, _ = grad_sink(b)
            if wbuf is not None:
                K.gemm_tn(dy2, x2, wbuf, M, ni, Kd, n, Kd, Kd, bbuf, A_ptr=dy2.data_ptr() + off * esz)
            off += ni
        return (dx.reshape(ctx.xshape) if dx is not None else None), None, None, None, None, None, None


# ---------------------------------------------------------------------------------------------------
# LayerNorm (stand-alone) with optional broadcast add (pos_embed)
# ---------------------------------------------------------------------------------------------------
class LayerNormFn(Function):
    """``alias=True``: also returns a second handle on ``x`` for the residual connection around the normalised branch; the
    gradient arriving on it is ADDED inside the LayerNorm-backward kernel (qavit_layernorm_bwd dres) instead of by an
    elementwise add of autograd's."""

    @staticmethod
    def forward(ctx, x, g, b, add, eps, act=False, alias=False):
        Cc = x.shape[-1]
        x2 = x.reshape(-1, Cc)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        add_rows = 0 if add is None else add.numel() // Cc
        K.layernorm_fwd(x2, y, g, b, eps, rows, Cc, mean, rstd, None if add is None else add.detach(), add_rows, act=act)
        ctx.save_for_backward(x2, g, b, add, mean, rstd)
        ctx.xshape = x.shape
        ctx.act = bool(act)
        ctx.alias = bool(alias)
        if alias:
            ctx.set_materialize_grads(False)
            return y.reshape(x.shape), x.view_as(x)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy, dalias=None):
        x2, g, b, add, mean, rstd = ctx.saved_tensors
        rows, Cc = x2.shape
        if dy is None:                                      # only the alias was differentiated through
            return dalias, None, None, None, None, None, None
        dy2 = dy.reshape(rows, Cc)
        if not dy2.is_contiguous():
            dy2 = dy2.contiguous()
        gbuf, _ = grad_sink(g)
        bbuf, _ = grad_sink(b)
        abuf, _ = grad_sink(add)
        dx = torch.empty_like(x2)
        DeferDW.arm()                                       # parameter gradients: partial sums now, one reduce launch when backward ends
        dres = None
        if dalias is not None:
            dres = dalias.reshape(rows, Cc)
            if not K.ln_dres_ok(x2, dres, Cc, abuf):
                dres = None
        K.layernorm_bwd(dy2, x2, g, mean, rstd, dx, gbuf, bbuf, rows, Cc, abuf, 0 if add is None else add.numel() // Cc,
                        beta=b.detach() if ctx.act else None, act=ctx.act, dres=dres)
        dx = dx.reshape(ctx.xshape)
        if dalias is not None and dres is None:
            dx = dx + dalias
        return dx, None, None, None, None, None, None


_DX_CAT = os.environ.get("QAVIT_DX_CAT", "1") != "0"
# MSDA's landmark path inside the fan node's GEMM: dk / dv of the pooled landmarks are scattered back to TOKEN rows (the pooling's backward
# applied to the 2C-wide dk | dv instead of to their C-wide product with Wkv: pooling and the projection commute) into columns [5C, 7C) of
# the group's matrix, ON A SIDE STREAM beside the next branch's backward kernel, and the fan GEMM contracts over 7C columns.  The landmark
# path's own input-gradient GEMM and the scatter behind it (two launches on the block's critical chain) are gone.
_MSDA_SIDE = os.environ.get("QAVIT_MSDA_SIDE", "1") != "0"
_MSDA_SIDE_STREAM = os.environ.get("QAVIT_MSDA_SIDE_STREAM", "0") != "0"


class FanGroup:
    """The fused attention branches that read ONE LayerNormFanFn output (norm1 of a QuadAttentionBlock: cross q_proj, SWA qkv, MSDA's
    q) each end their backward in an input-gradient GEMM of the same [M, C] shape whose results the fan node then sums.  Instead the
    branch backward kernels write dq (| dk | dv) into column slices of ONE [M, 5C] matrix and return no gradient; the fan node's backward
    runs ONE GEMM over the concatenated contraction axis against the row-stack [Wq_cross; Wqkv_swa; Wqkv_msda] (its transposed pack,
    K = the first 5C columns) -- the sum of the three products is what the contraction computes -- with the landmark-path gradient of MSDA
    as its residual addend (dx = dq_c Wq_c + [dq dk dv]_s Wqkv_s + dq_m Wq_m + dx_pool; HQAViT_CIFAR100.py:448, :523, :613 backward).
    Two launches, two [M, C] round trips and two addends of the fan-in sum fewer per block."""
    registry = {}                                           # data_ptr of the fan-out tensor -> weakref(FanGroup)
    _side = {}                                              # device index -> the stream MSDA's landmark scatter runs on

    def __init__(self, M, Cc, dtype, device):
        self.M, self.C, self.dtype, self.device = M, Cc, dtype, device
        self.ld = 7 * Cc                                    # cross q | SWA q k v | MSDA q | MSDA k v (token rows, _MSDA_SIDE)
        self.buf = None
        self.entries = {}                                   # kind -> weight parameter
        self.resid = None
        self.kv = False                                     # columns [5C, 7C) hold MSDA's dk | dv scattered back to token rows
        self.kv_side = None                                 # ... written on this stream: run() waits for it

    def _alloc(self):
        if self.buf is None:
            self.buf = torch.empty(self.M, 7 * self.C, dtype=self.dtype, device=self.device)
        return self.buf

    def kv_ptr(self):
        return self._alloc().data_ptr() + 5 * self.C * self.buf.element_size()

    @classmethod
    def side_stream(cls, device):
        key = torch.device(device).index or 0
        st = cls._side.get(key)
        if st is None:
            st = cls._side[key] = torch.cuda.Stream(device=device)
        return st

    # column offset / width of a branch kind's slice: cross q | SWA q k v | MSDA q
    def slot(self, kind):
        Cc = self.C
        return {2: (0, Cc), 0: (Cc, 3 * Cc), 1: (4 * Cc, Cc)}[kind]

    def slice_ptr(self, kind):
        return self._alloc().data_ptr() + self.slot(kind)[0] * self.buf.element_size()

    def kdim(self):
        return (7 if self.kv else 5) * self.C

    @classmethod
    def create(cls, y, rows, Cc):
        if not (_DX_CAT and y.is_cuda and y.dtype == torch.bfloat16 and rows >= 16 and Cc % 64 == 0):
            return None
        grp = cls(rows, Cc, y.dtype, y.device)
        if len(cls.registry) > 256:                         # addresses of groups long gone (an eager run over many shapes): drop the dead references
            for key in [k_ for k_, r in cls.registry.items() if r() is None]:
                del cls.registry[key]
        cls.registry[y.data_ptr()] = weakref.ref(grp)
        return grp

    @classmethod
    def lookup(cls, x):
        """The live group whose fan-out tensor ``x`` is a view of (called in a branch's FORWARD, while that tensor is alive)."""
        ref = cls.registry.get(x.data_ptr())
        grp = ref() if ref is not None else None
        if grp is None or grp.M != x.numel() // x.shape[-1] or grp.C != x.shape[-1] or grp.dtype != x.dtype:
            return None
        return grp

    def complete(self) -> bool:
        return all(k in self.entries for k in (2, 0, 1))

    def run(self, lnbwd=None, dres=None):
        """-> the summed input gradient [M, C] of the registered branches, or None.  ``lnbwd`` (all three branches registered only): the
        fan node's LayerNorm backward as the GEMM's epilogue -- the result is then the gradient of the LayerNorm's INPUT, with ``dres``
        (the residual path's gradient) added behind it and the landmark-path gradient in front of it."""
        if not self.entries:
            return None
        M, Cc = self.M, self.C
        out = torch.empty(M, Cc, dtype=self.dtype, device=self.device)
        order = (2, 0, 1)
        ld = self.ld
        esz = self.buf.element_size()
        if self.kv_side is not None:                        # the landmark scatter ran beside the branch kernels: join it here
            torch.cuda.current_stream(self.device).wait_stream(self.kv_side)
            self.kv_side = None
        if self.complete():
            _, Wt = pack_for(self.device).get([self.entries[2], self.entries[0], self.entries[1]], self.dtype)      # [C, C + 3C + 3C]
            Kc = self.kdim()
            if lnbwd is not None:
                lnbwd = dict(lnbwd, adds=[self.resid] + list(lnbwd.get("adds") or ()))
                K.gemm_nt(self.buf, Wt, out, M, Cc, Kc, ld, Wt.shape[1], Cc, None, R=dres, ldr=Cc, lnbwd=lnbwd)
            else:
                K.gemm_nt(self.buf, Wt, out, M, Cc, Kc, ld, Wt.shape[1], Cc, None, R=self.resid, ldr=Cc)
            return out
        assert lnbwd is None
        first = True                                        # a branch without a gradient this pass: a GEMM per registered slice
        for k in order:
            if k not in self.entries:
                continue
            off, width = self.slot(k)
            if k == 1 and self.kv:
                width = 3 * Cc                              # q | k | v of MSDA are adjacent columns, as in its transposed weight pack
            _, Wt = pack_for(self.device).get(self.entries[k], self.dtype)
            R = self.resid if first else out
            K.gemm_nt(self.buf, Wt, out, M, Cc, width, ld, Wt.shape[1], Cc, None, R=R, ldr=Cc, A_ptr=self.buf.data_ptr() + off * esz)
            first = False
        if first:
            return None
        return out


class LayerNormFanFn(Function):
    """LayerNorm whose output feeds ``k`` consumers, plus the alias of ``x`` for the residual: -> (y_1 .. y_k, x_alias).  Backward gets
    the k gradients and the residual's and runs ONE launch: the LayerNorm-backward kernel sums them on load (qavit_layernorm_bwd_sum)
    instead of a k-way sum kernel in front of it (FanOutFn)."""

    @staticmethod
    def forward(ctx, x, g, b, eps, k):
        Cc = x.shape[-1]
        x2 = x.reshape(-1, Cc)
        if not x2.is_contiguous():
            x2 = x2.contiguous()
        rows = x2.shape[0]
        y = torch.empty_like(x2)
        mean = torch.empty(rows, dtype=torch.float32, device=x.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
        K.layernorm_fwd(x2, y, g, b, eps, rows, Cc, mean, rstd, None, 0, act=False)
        ctx.save_for_backward(x2, g, b, mean, rstd)
        ctx.xshape, ctx.k = x.shape, k
        ctx.set_materialize_grads(False)
        # the consumers' qkv input-gradient GEMMs, deferred to this node's backward (grad mode is off inside Function.forward: ask the ctx)
        ctx.group = FanGroup.create(y, rows, Cc) if ctx.needs_input_grad[0] else None
        y = y.reshape(x.shape)
        return tuple(y.view_as(y) for _ in range(k)) + (x.view_as(x),)

    @staticmethod
    def backward(ctx, *grads):
        x2, g, b, mean, rstd = ctx.saved_tensors
        rows, Cc = x2.shape
        dalias = grads[ctx.k]
        dys = [t.reshape(rows, Cc) for t in grads[:ctx.k] if t is not None]
        dys = [t if t.is_contiguous() else t.contiguous() for t in dys]
        grp = getattr(ctx, "group", None)
        ctx.group = None
        dres = None
        if dalias is not None:
            dres = dalias.reshape(rows, Cc)
            if not K.ln_dres_ok(x2, dres, Cc):
                dres = None
        if grp is not None and grp.complete() and _LN_EPI and (dalias is None or dres is not None) and \
                len(dys) + (grp.resid is not None) <= 2 and K.gemm_nt_lnbwd_ok(x2, rows, Cc, grp.kdim(), 0, grp.resid, dres, *dys):
            # ONE launch for the whole node: the branches' input-gradient GEMM (K = 5C) with this LayerNorm's backward as its epilogue
            # -- the remaining consumers' gradients (the channel-group branch) and MSDA's landmark-path gradient are added to the
            # product on load, the residual path's gradient behind the LayerNorm backward
            gbuf, _ = grad_sink(g)
            bbuf, _ = grad_sink(b)
            DeferDW.arm()
            dx = grp.run(lnbwd=dict(x=x2, mean=mean, rstd=rstd, gamma=g, dgamma=gbuf, dbeta=bbuf, adds=dys), dres=dres)
            return dx.reshape(ctx.xshape), None, None, None, None
        if grp is not None:
            dcat = grp.run()                                # ONE GEMM for the branches that left their dq / dk / dv in the group's matrix
            if dcat is not None:
                dys.insert(0, dcat)
        if not dys:
            return dalias, None, None, None, None
        gbuf, _ = grad_sink(g)
        bbuf, _ = grad_sink(b)
        dx = torch.empty_like(x2)
        DeferDW.arm()
        if K.layernorm_bwd_sum_ok(x2, dys, Cc):
            K.layernorm_bwd_sum(dys, x2, g, mean, rstd, dx, gbuf, bbuf, rows, Cc, dres=dres)
        else:
            dy = dys[0]
            for t in dys[1:]:
                dy = dy + t
            K.layernorm_bwd(dy, x2, g, mean, rstd, dx, gbuf, bbuf, rows, Cc, dres=dres)
        dx = dx.reshape(ctx.xshape)
        if dalias is not None and dres is None:
            dx = dx + dalias
        return dx, None, None, None, None


def layer_norm(x, g, b, eps=1e-5, add=None, act=None, alias=False):
    """LayerNorm (+ pos_embed-style broadcast add); ``act="gelu"`` fuses the exact GELU that follows it.  ``alias=True`` -> (y,
    x_alias): use ``x_alias`` for the residual connection (see LayerNormFn)."""
    if alias and torch.is_grad_enabled() and x.requires_grad:
        return LayerNormFn.apply(x, g, b, add, eps, act == "gelu", True)
    y = LayerNormFn.apply(x, g, b, add, eps, act == "gelu")
    return (y, x) if alias else y


# ---------------------------------------------------------------------------------------------------
# attention core
# ---------------------------------------------------------------------------------------------------
def _attn_drop(a, spec, rt):
    p, site = spec.get("drop", (0.0, 0))
    if p > 0.0:
        a.drop_p, a.drop_site, a.rng = float(p), int(site), rt.rng.data_ptr()


def _defer_fix(rt, a, out, bias, proj_drop, trip, o, ldos, Co):
    """The caller writes the bank from ``out`` next (modules._Branch._write): the NaN rule's rewrite launch is skipped and its parameters
    wait in rt.pending_fix for bank_write, whose statistics kernel does the rewrite for the images it visits before reading them."""
    if rt.pending_fix is not None:
        raise RuntimeError("a deferred NaN rule was never consumed by a bank write")
    a.nan_defer = 1
    fx = L.NanFix()
    fx.flag, fx.trip, fx.bias = rt.nan_flag.data_ptr(), K._p(trip), bias.data_ptr()
    fx.drop_p, fx.drop_site = float(proj_drop[0]), int(proj_drop[1])
    fx.rng = rt.rng.data_ptr() if proj_drop[0] > 0.0 else None
    fx.o_save, fx.ldos, fx.Co = K._p(o), ldos, Co
    rt.pending_fix = (fx, out, (bias, trip, o))


_BRANCH_DRAIN = os.environ.get("QAVIT_BRANCH_DRAIN", "0") != "0"


def branch_forward(kind, x, wqkv, bqkv, wproj, bproj, E_k, E_v, sh_k, sh_v, pool_idx=None, pool_stride=0, Lk=0,
                   attn_drop=(0.0, 0), proj_drop=(0.0, 0), want_o=False, save=False, defer_fix=False):
    """One launch for a whole attention branch on 16- or 64-token problems (csrc/branch_fwd.hip; include/qavit.h qavit_branch_args):
    ``kind`` 0 = SWA, 1 = MSDA, 2 = cross.  ``x`` [B, 16 | 64, 192] bf16 (norm1's output); ``wqkv`` / ``wproj`` are the fp32
    parameters (read through the fragment-packed copies of the WeightPack); ``sh_k`` / ``sh_v`` fp32 [16, 192] (the bank, or
    its k_proj / v_proj for cross).  No autograd: the caller owns the backward.  -> out [B, 16, 192] (and O when ``want_o``)."""
    K._require_cuda(x)
    rt = _rt(x)
    B, T, Cc = x.shape
    x2 = x.reshape(B * T, Cc)
    if not x2.is_contiguous():
        x2 = x2.contiguous()
    pack = pack_for(x.device)
    a = L.BranchArgs()
    a.dtype, a.kind = K.dt_code(x.dtype), kind
    a.drain_waits = 1 if _BRANCH_DRAIN else 0               # diagnostic: the ring's counted waits replaced by full drains (same bits)
    a.B, a.T, a.C, a.H, a.D = B, T, Cc, 4, Cc // 4
    a.KC, a.S, a.L = (E_k.shape[1] if E_k is not None else 0), sh_k.shape[-2], Lk
    a.x, a.ldx = x2.data_ptr(), Cc
    a.wqkv_frag, a.bqkv = pack.get_frag(wqkv).data_ptr(), bqkv.data_ptr()
    a.wproj_frag, a.bproj = pack.get_frag(wproj).data_ptr(), bproj.data_ptr()
    if E_k is not None:
        a.E_k, a.E_v = E_k.data_ptr(), E_v.data_ptr()
    a.sh_k, a.sh_v = sh_k.data_ptr(), sh_v.data_ptr()
    if pool_idx is not None:
        a.pool_idx, a.pool_stride = pool_idx.data_ptr(), pool_stride
    out = torch.empty(B * T, Cc, dtype=x.dtype, device=x.device)
    a.out, a.ldo = out.data_ptr(), Cc
    o = None
    if want_o:
        o = torch.empty_like(out)
        a.o_save = o.data_ptr()
    saved = None
    if save and rt.nan_guard:                              # did the NaN rule fire?  (read by the fused backward: zero gradient through the core)
        trip = torch.empty(1, dtype=torch.int32, device=x.device)
        a.nan_trip = trip.data_ptr()
    else:
        trip = None
    if save:                                               # q / k / v (and MSDA's landmarks) for the backward pass, written by the kernel
        esz = x.element_size()
        if kind == 0:
            qkv = torch.empty(B * T, 3 * Cc, dtype=x.dtype, device=x.device)
            a.q_save, a.ldq_save = qkv.data_ptr(), 3 * Cc
            a.kv_save, a.ldkv_save = qkv.data_ptr() + Cc * esz, 3 * Cc
            saved = (qkv,)
        elif kind == 1:
            q = torch.empty(B * T, Cc, dtype=x.dtype, device=x.device)
            kv = torch.empty(B * Lk, 2 * Cc, dtype=x.dtype, device=x.device)
            pooled = torch.empty(B * Lk, Cc, dtype=x.dtype, device=x.device)
            a.q_save, a.ldq_save = q.data_ptr(), Cc
            a.kv_save, a.ldkv_save = kv.data_ptr(), 2 * Cc
            a.pooled_save = pooled.data_ptr()
            saved = (q, kv, pooled)
        else:
            q = torch.empty(B * T, Cc, dtype=x.dtype, device=x.device)
            a.q_save, a.ldq_save = q.data_ptr(), Cc
            saved = (q,)
    a.attn_drop_p, a.attn_drop_site = float(attn_drop[0]), int(attn_drop[1])
    a.proj_drop_p, a.proj_drop_site = float(proj_drop[0]), int(proj_drop[1])
    if attn_drop[0] > 0.0 or proj_drop[0] > 0.0:
        a.rng = rt.rng.data_ptr()
    if rt.nan_guard:
        a.nan_flag = rt.nan_flag.data_ptr()
        if defer_fix and _DEFER_FIX:
            _defer_fix(rt, a, out, bproj, proj_drop, trip, o, Cc, Cc)
    K.branch_fwd(a)
    out = out.reshape(B, T, Cc)
    if save:
        return out, o, saved, trip
    return (out, o) if want_o else out


class BranchFn(Function):
    """A whole attention branch on 16-token problems as ONE forward launch (csrc/branch_fwd.hip).  ``meta``: kind (0 SWA,
    1 MSDA, 2 cross), pool_idx / pool_stride / Lk (MSDA), attn_drop / proj_drop = (p, site).
    Backward runs the unfused kernels (proj dX / dW, attention-core backward, qkv dX / dW) on what the forward kernel saved:
    q / k / v (k and v through a second, transposed MFMA on the same fragments -- cheaper than a recompute GEMM launch per
    branch), MSDA's pooled landmarks and the attention output O.  Dropout masks are pure functions of (seed, step, site,
    element), identical in the fused forward and the unfused backward kernels."""

    @staticmethod
    def forward(ctx, x, wqkv, bqkv, wproj, bproj, E_k, E_v, sh_k, sh_v, meta):
        need = any(ctx.needs_input_grad)
        res = branch_forward(meta["kind"], x, wqkv, bqkv, wproj, bproj, E_k, E_v, sh_k.reshape(-1, x.shape[-1]), sh_v.reshape(-1, x.shape[-1]),
                             meta.get("pool_idx"), meta.get("pool_stride", 0), meta.get("Lk", 16 if meta["kind"] == 0 else 0), meta["attn_drop"], meta["proj_drop"],
                             want_o=need, save=need, defer_fix=bool(meta.get("defer_fix")))
        if not need:
            return res
        out, o, saved, trip = res
        ctx.n_saved = len(saved)
        ctx.trip = trip
        # The bank is mutated in place later in the same forward (GlobalTokenBank.write); the reference's SDPA backward sees
        # the values its forward used (torch.cat made a copy), so snapshot shared rows that alias a parameter (as AttnFn does).
        sk_s, sv_s = sh_k, sh_v
        if sh_k.is_leaf and sh_v.is_leaf:
            snap = meta.get("bank_snap")                    # the copy the previous bank write left (qavit_bank_apply snap_*), else copy now
            sk_s, sv_s = snap if snap is not None else K.copy2(sh_k, sh_v)
        ctx.sh_alias = (sh_k, sh_v)
        ctx.meta = meta
        ctx.fan = FanGroup.lookup(x) if ctx.needs_input_grad[0] else None      # x is one output of a LayerNormFanFn: see FanGroup
        ctx.save_for_backward(x, wqkv, bqkv, wproj, bproj, E_k, E_v, sk_s, sv_s, o, *saved)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wqkv, bqkv, wproj, bproj, E_k, E_v, sk_s, sv_s, o = ctx.saved_tensors[:10]
        saved = ctx.saved_tensors[10:]
        sh_k_in, sh_v_in = ctx.sh_alias
        m = ctx.meta
        kind = m["kind"]
        B, T, Cc = x.shape
        H, D = 4, Cc // 4
        S = sk_s.reshape(-1, Cc).shape[0]
        x2 = x.reshape(B * T, Cc)
        if _BRANCH_BWD and dout.dtype == torch.bfloat16 and x.dtype == torch.bfloat16:
            return _branch_backward_fused(ctx, dout, x, x2, wqkv, bqkv, wproj, bproj, E_k, E_v, sk_s, sv_s, o, saved, sh_k_in, sh_v_in, m, B, T, Cc, S)
        # ---- proj backward: dO = (dout * mask) Wproj ; dWproj += (dout * mask)^T O ; dbproj
        d_o = _linear_bwd(o, wproj, bproj, dout.reshape(B * T, Cc), 0, Cc, True, drop=m["proj_drop"])
        # ---- the attention-core backward on the saved projections, then the projections' own backward
        with torch.no_grad():
            if kind == 0:
                qkv = saved[0]
                nwin = T // 16                              # 64 tokens: the four 4x4 windows through the row table
                spec = dict(mode=0, G=B * nwin, Nq=16, L=16, H=H, D=D, KC=E_k.shape[1], S=S, groups_per_b=nwin, q_rows_per_b=T, k_rows_per_b=T,
                            q_tbl=m.get("win_tbl") if nwin > 1 else None, k_tbl=m.get("win_tbl") if nwin > 1 else None,
                            q_off=0, k_off=Cc, v_off=2 * Cc, q_rows=B * T, drop=m["attn_drop"])
                dq_t, _, ek_ret, ev_ret, sk_ret, sv_ret = _attn_bwd(qkv, None, E_k, E_v, sk_s, sv_s, sh_k_in, sh_v_in, spec, d_o)
                dx = _linear_bwd(x2, wqkv, bqkv, dq_t, 0, 3 * Cc, ctx.needs_input_grad[0])
            elif kind == 1:
                idx, stride, Lk = m["pool_idx"], m["pool_stride"], m["Lk"]
                NP = idx.numel() // stride
                q, kv, p2 = saved
                spec = dict(mode=0, G=B, Nq=T, L=Lk, H=H, D=D, KC=E_k.shape[1], S=S, groups_per_b=1, q_rows_per_b=T, k_rows_per_b=NP,
                            q_off=0, k_off=0, v_off=Cc, q_rows=B * T, drop=m["attn_drop"])
                dq_t, dkv_t, ek_ret, ev_ret, sk_ret, sv_ret = _attn_bwd(q, kv, E_k, E_v, sk_s, sv_s, sh_k_in, sh_v_in, spec, d_o)
                dpool = _linear_bwd(p2, wqkv, bqkv, dkv_t, Cc, 2 * Cc, ctx.needs_input_grad[0])
                dxp = None
                if ctx.needs_input_grad[0]:
                    dxp = torch.empty(B * T, Cc, dtype=x.dtype, device=x.device)
                    K.gather_pool_bwd(dpool.reshape(B, NP, Cc).contiguous(), idx, dxp, B, T, NP, stride, Cc)
                dx = _linear_bwd(x2, wqkv, bqkv, dq_t, 0, Cc, ctx.needs_input_grad[0], dx_add=dxp)     # dq Wq + (landmark-path gradient)
            else:
                q = saved[0]
                spec = dict(mode=1, G=B, Nq=T, L=0, H=H, D=D, S=S, q_off=0, k_off=0, v_off=0, q_rows=B * T, drop=m["attn_drop"])
                dq_t, _, ek_ret, ev_ret, sk_ret, sv_ret = _attn_bwd(q, None, None, None, sk_s.reshape(S, Cc), sv_s.reshape(S, Cc),
                                                                    sh_k_in, sh_v_in, spec, d_o)
                dx = _linear_bwd(x2, wqkv, bqkv, dq_t, 0, Cc, ctx.needs_input_grad[0])
        gE_k = _ret(ek_ret, E_k) if ek_ret is not None else None
        gE_v = _ret(ev_ret, E_v) if ev_ret is not None else None
        return (dx.reshape(B, T, Cc) if dx is not None else None), None, None, None, None, gE_k, gE_v, \
            _ret(sk_ret, sh_k_in), _ret(sv_ret, sh_v_in), None


_BRANCH_BWD = os.environ.get("QAVIT_FUSED_BRANCH_BWD", "1") != "0"


def _branch_backward_fused(ctx, dout, x, x2, wqkv, bqkv, wproj, bproj, E_k, E_v, sk_s, sv_s, o, saved, sh_k_in, sh_v_in, m, B, T, Cc, S):
    """BranchFn.backward through csrc/branch_bwd.hip: ONE launch for the proj input gradient + the attention-core backward, then
    the qkv input-gradient GEMM(s); weight gradients deferred as everywhere, the sums over images (dE_k, dE_v, shared rows) left as
    per-workgroup partial rows for the end-of-backward reduce launch (at once where a consumer waits: cross-attention's projected bank)."""
    kind = m["kind"]
    rt = _rt(x)
    M = B * T
    g2 = dout.reshape(M, Cc)
    if not g2.is_contiguous():
        g2 = g2.contiguous()
    pd, ad = m["proj_drop"], m["attn_drop"]
    dz = torch.empty_like(g2) if pd[0] > 0.0 else g2
    a = L.BranchBwdArgs()
    a.dtype, a.kind = K.dt_code(x.dtype), kind
    a.B, a.T, a.C, a.H, a.D, a.S = B, T, Cc, 4, Cc // 4, S
    a.dout, a.lddout = g2.data_ptr(), Cc
    a.wprojT_frag = pack_for(x.device).get_frag(wproj, x.dtype, order=3).data_ptr()
    a.o, a.ldo = o.data_ptr(), Cc
    a.sh_k, a.sh_v = sk_s.data_ptr(), sv_s.data_ptr()
    a.attn_drop_p, a.attn_drop_site = float(ad[0]), int(ad[1])
    a.proj_drop_p, a.proj_drop_site = float(pd[0]), int(pd[1])
    a.rng = rt.rng.data_ptr()
    if pd[0] > 0.0:
        a.dz, a.lddz = dz.data_ptr(), Cc
    if getattr(ctx, "trip", None) is not None:
        a.nan_trip = ctx.trip.data_ptr()
    esz = 2
    dkv = None
    # the qkv input-gradient GEMM deferred to the fan node that produced x (FanGroup): dq (| dk | dv) go into its [M, 5C] matrix
    grp = getattr(ctx, "fan", None) if (ctx.needs_input_grad[0] and wqkv.requires_grad) else None
    if grp is not None and (grp.M != M or grp.C != Cc or kind in grp.entries):
        grp = None
    dq_ptr, dq_ld = None, None
    if grp is not None:
        dq_ptr, dq_ld = grp.slice_ptr(kind), grp.ld
    if kind == 0:
        qkv = saved[0]
        dq = torch.empty_like(qkv) if grp is None else None
        a.q, a.ldq = qkv.data_ptr(), 3 * Cc
        a.k_tok, a.v_tok, a.ldkv, a.kv_rows = qkv.data_ptr() + Cc * esz, qkv.data_ptr() + 2 * Cc * esz, 3 * Cc, 16
        if grp is None:
            dq_ptr, dq_ld = dq.data_ptr(), 3 * Cc
        a.dq, a.lddq = dq_ptr, dq_ld
        a.dk_tok, a.dv_tok, a.lddkv = dq_ptr + Cc * esz, dq_ptr + 2 * Cc * esz, dq_ld
        a.KC, a.L = E_k.shape[1], 16
    elif kind == 1:
        q, kv, p2 = saved
        Lk = m["Lk"]
        dq, dkv = (torch.empty_like(q) if grp is None else None), torch.empty_like(kv)
        a.q, a.ldq = q.data_ptr(), Cc
        a.k_tok, a.v_tok, a.ldkv, a.kv_rows = kv.data_ptr(), kv.data_ptr() + Cc * esz, 2 * Cc, Lk
        if grp is None:
            dq_ptr, dq_ld = dq.data_ptr(), Cc
        a.dq, a.lddq = dq_ptr, dq_ld
        a.dk_tok, a.dv_tok, a.lddkv = dkv.data_ptr(), dkv.data_ptr() + Cc * esz, 2 * Cc
        a.KC, a.L = E_k.shape[1], Lk
    else:
        q = saved[0]
        dq = torch.empty_like(q) if grp is None else None
        a.q, a.ldq = q.data_ptr(), Cc
        if grp is None:
            dq_ptr, dq_ld = dq.data_ptr(), Cc
        a.dq, a.lddq = dq_ptr, dq_ld
    if kind != 2:
        a.E_k, a.E_v = E_k.data_ptr(), E_v.data_ptr()
    nparts = K.branch_bwd_parts(B, T)
    PF = K.BRANCH_PARTS_FLOATS_64 if T == 64 else K.BRANCH_PARTS_FLOATS
    PE = (48 if T == 64 else 16) * 32                      # floats per dE slot of a partial row: [dE_k | dE_v | d sh_k | d sh_v]
    parts = torch.empty(nparts * PF, dtype=torch.float32, device=x.device)
    a.parts, a.parts_stride = parts.data_ptr(), PF
    DeferDW.arm()
    K.branch_bwd(a)
    # ---- dW_proj += dz^T O, db_proj += colsum(dz)
    wbuf, _ = grad_sink(wproj)
    bbuf, _ = grad_sink(bproj)
    if wbuf is not None or bbuf is not None:
        if wbuf is None:
            wbuf = torch.zeros(wproj.shape, dtype=torch.float32, device=x.device)
        K.gemm_tn(dz, o, wbuf, M, Cc, Cc, Cc, Cc, Cc, bbuf)
    # ---- the partial rows: [dE_k | dE_v | d sh_k | d sh_v]
    ek_ret = ev_ret = None
    if kind != 2:
        ek_buf, ek_ret = grad_sink(E_k)
        ev_buf, ev_ret = grad_sink(E_v)
        if ek_buf is not None or ev_buf is not None:
            ce = min(PE, E_k.numel())                       # SWA: the 16 window rows; MSDA: the slot (E has 128 rows, the first L <= 16 | 48 are used)
            if ce == PE:
                K.DeferredLN.push_raw(parts.data_ptr(), nparts, ce, K._p(ek_buf), K._p(ev_buf), PF, (parts, ek_buf, ev_buf))
            else:
                K.DeferredLN.push_raw(parts.data_ptr(), nparts, ce, K._p(ek_buf), None, PF, (parts, ek_buf))
                K.DeferredLN.push_raw(parts.data_ptr() + PE * 4, nparts, ce, K._p(ev_buf), None, PF, (parts, ev_buf))
    leaf = sh_k_in.is_leaf and sh_v_in.is_leaf
    rec = None if leaf else DeferredBank.record_of(sh_k_in, sh_v_in)
    if leaf:
        sk_buf, sk_ret = grad_sink(sh_k_in)
        sv_buf, sv_ret = grad_sink(sh_v_in)
    elif rec is not None:                                  # BankProj2Fn's rows: their projection's backward runs once, when the pass ends
        both = DeferredBank.slot(sh_k_in.shape, x.device)
        sk_buf, sv_buf = both[0], both[1]
        sk_ret = sv_ret = None
    else:                                                  # a consumer waits for these (the bank projections' backward): reduce now
        both = torch.zeros((2,) + tuple(sh_k_in.shape), dtype=torch.float32, device=x.device)
        sk_buf = sk_ret = both[0] if sh_k_in.requires_grad else None
        sv_buf = sv_ret = both[1] if sh_v_in.requires_grad else None
    half = (S * Cc) // 2
    descs = []
    for buf, off in ((sk_buf, 2 * PE), (sv_buf, 2 * PE + S * Cc)):
        if buf is not None:
            descs.append((parts.data_ptr() + off * 4, nparts, half, buf.data_ptr(), buf.data_ptr() + half * 4, PF))
    if descs:
        if leaf or rec is not None:
            for d_ in descs:
                K.DeferredLN.push_raw(*d_, (parts, sk_buf, sv_buf))
            if rec is not None:
                DeferredBank.queue.append((rec, both))
        else:
            K.reduce_now([K.DeferredLN.desc(*d_) for d_ in descs])
    # ---- the projections' own backward
    need_dx = ctx.needs_input_grad[0]
    with torch.no_grad():
        if grp is not None:
            # input gradient: the fan node's ONE GEMM over every registered slice; here only the weight / bias gradients, whose dz operand
            # is this branch's column slice of the group's matrix (leading dimension 5C)
            nq = 3 * Cc if kind == 0 else Cc
            dxp = None
            if kind == 1:
                idx, stride = m["pool_idx"], m["pool_stride"]
                NP = idx.numel() // stride
                if _MSDA_SIDE and dkv.is_contiguous() and dkv.data_ptr() % 16 == 0 and (2 * Cc) % 8 == 0:
                    _linear_bwd(saved[2], wqkv, bqkv, dkv, Cc, 2 * Cc, False)        # weight / bias gradients only
                    kvp = grp.kv_ptr()
                    if _MSDA_SIDE_STREAM:
                        side, cur = FanGroup.side_stream(x.device), torch.cuda.current_stream(x.device)
                        side.wait_stream(cur)               # dk | dv of the landmarks exist
                        dkv.record_stream(side)
                        with torch.cuda.stream(side):
                            K.gather_pool_bwd_ld(dkv, idx, kvp, grp.ld, B, T, NP, stride, 2 * Cc)
                        grp.kv, grp.kv_side = True, side
                    else:
                        K.gather_pool_bwd_ld(dkv, idx, kvp, grp.ld, B, T, NP, stride, 2 * Cc)
                        grp.kv = True
                else:
                    dpool = _linear_bwd(saved[2], wqkv, bqkv, dkv, Cc, 2 * Cc, need_dx)
                    dxp = torch.empty(M, Cc, dtype=x.dtype, device=x.device)
                    K.gather_pool_bwd(dpool.reshape(B, NP, Cc).contiguous(), idx, dxp, B, T, NP, stride, Cc)
                    grp.resid = dxp
            wq_buf, _ = grad_sink(wqkv)
            bq_buf, _ = grad_sink(bqkv)
            K.gemm_tn(grp.buf, x2, wq_buf, M, nq, Cc, grp.ld, Cc, Cc, None, A_ptr=dq_ptr, C_ptr=wq_buf.data_ptr(),
                      colsum_ptr=None if bq_buf is None else bq_buf.data_ptr())
            grp.entries[kind] = wqkv
            dx = None
        elif kind == 0:
            dx = _linear_bwd(x2, wqkv, bqkv, dq, 0, 3 * Cc, need_dx)
        elif kind == 1:
            idx, stride = m["pool_idx"], m["pool_stride"]
            NP = idx.numel() // stride
            dpool = _linear_bwd(saved[2], wqkv, bqkv, dkv, Cc, 2 * Cc, need_dx)
            dxp = None
            if need_dx:
                dxp = torch.empty(M, Cc, dtype=x.dtype, device=x.device)
                K.gather_pool_bwd(dpool.reshape(B, NP, Cc).contiguous(), idx, dxp, B, T, NP, stride, Cc)
            dx = _linear_bwd(x2, wqkv, bqkv, dq, 0, Cc, need_dx, dx_add=dxp)
        else:
            dx = _linear_bwd(x2, wqkv, bqkv, dq, 0, Cc, need_dx)
    gE_k = _ret(ek_ret, E_k) if ek_ret is not None else None
    gE_v = _ret(ev_ret, E_v) if ev_ret is not None else None
    return (dx.reshape(B, T, Cc) if dx is not None else None), None, None, None, None, gE_k, gE_v, \
        _ret(sk_ret, sh_k_in) if sk_ret is not None else None, _ret(sv_ret, sh_v_in) if sv_ret is not None else None, None


def branch_ok(kind, x, Lk, KC, S, heads) -> bool:
    """Does the fused branch kernel cover this call?  (bf16, 16 or 64 tokens x 192 channels, 4 heads of 48, 16 shared rows, KC = 32)"""
    if x.dtype != torch.bfloat16 or x.dim() != 3 or not x.is_cuda or os.environ.get("QAVIT_FUSED_BRANCH", "1") == "0":
        return False
    B, T, Cc = x.shape
    if heads == 0 or Cc % heads:
        return False
    return K.branch_supported(kind, T, Cc, heads, Cc // heads, KC, S, Lk)


_CGA_FUSED = os.environ.get("QAVIT_FUSED_CGA", "1") != "0"
_CFUSE = os.environ.get("QAVIT_FUSED_COMPRESS", "1") != "0"
_CFUSE_BWD = os.environ.get("QAVIT_FUSED_COMPRESS_BWD", "1") != "0"
_CGA_FUSED_BWD = os.environ.get("QAVIT_FUSED_CGA_BWD", "1") != "0"


def cga_ok(x, G, heads, S) -> bool:
    if not _CGA_FUSED or x.dtype != torch.bfloat16 or x.dim() != 3 or not x.is_cuda:
        return False
    B, T, Cc = x.shape
    return bool(L.load().qavit_cga_supported(T, Cc, G, heads, S))


class CGABranchFn(Function):
    """The whole channel-group branch (HQAViT_CIFAR100.py:535-595) in one forward launch (csrc/cga.hip): q/k/v projections of the six
    channel groups, 4 heads of D = 4 over [tokens ; projected bank rows], softmax + dropout, P.V, proj + dropout.  Backward: the
    unfused kernels on q/k/v recomputed by one GEMM (they are 9 MB the forward does not store) and the saved attention output."""

    @staticmethod
    def forward(ctx, x, wq, bq, wk, bk, wv, bv, wproj, bproj, sh_k, sh_v, meta):
        K._require_cuda(x, wq)
        rt = _rt(x)
        B, T, Cc = x.shape
        G, H, S = meta["G"], meta["H"], sh_k.shape[0]
        x = x.contiguous()
        pk = pack_for(x.device)
        Wqkv, _ = pk.get([wq, wk, wv], x.dtype)
        bqkv = pk.get([bq, bk, bv], torch.float32)[0].reshape(-1)
        Wp, _ = pk.get(wproj, x.dtype)
        need = any(ctx.needs_input_grad)
        out = torch.empty(B, T, Cc, dtype=x.dtype, device=x.device)
        o = torch.empty(B * T, wproj.shape[1], dtype=x.dtype, device=x.device) if need else None
        a = L.CgaArgs()
        a.dtype = K.dt_code(x.dtype)
        a.B, a.T, a.C, a.G, a.H, a.D, a.S = B, T, Cc, G, H, (wq.shape[0] // H), S
        a.x, a.ldx = x.data_ptr(), Cc
        a.wqkv_rm, a.bqkv = Wqkv.data_ptr(), bqkv.data_ptr()
        a.wproj_rm, a.bproj = Wp.data_ptr(), bproj.data_ptr()
        shk, shv = sh_k.detach().contiguous(), sh_v.detach().contiguous()
        a.sh_k, a.sh_v = shk.data_ptr(), shv.data_ptr()
        a.out, a.ldo = out.data_ptr(), Cc
        a.o_save = K._p(o)
        ad, pd = meta["attn_drop"], meta["proj_drop"]
        a.attn_drop_p, a.attn_drop_site = float(ad[0]), int(ad[1])
        a.proj_drop_p, a.proj_drop_site = float(pd[0]), int(pd[1])
        a.rng = rt.rng.data_ptr()
        trip = None
        if rt.nan_guard:
            a.nan_flag = rt.nan_flag.data_ptr()
            if need:
                trip = torch.empty(1, dtype=torch.int32, device=x.device)
                a.nan_trip = trip.data_ptr()
            if meta.get("defer_fix") and _DEFER_FIX:
                co = o.shape[-1] if o is not None else 0
                _defer_fix(rt, a, out, bproj, pd, trip, o, co, co)
        L.check(L.load().qavit_cga_fwd(C.byref(a), K.stream()), "cga_fwd")
        if need:
            ctx.trip = trip
            ctx.meta = meta
            ctx.save_for_backward(x, wq, bq, wk, bk, wv, bv, wproj, bproj, shk, shv, o)
            ctx.sh_in = (sh_k, sh_v)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, wq, bq, wk, bk, wv, bv, wproj, bproj, shk, shv, o = ctx.saved_tensors
        sh_k_in, sh_v_in = ctx.sh_in
        m = ctx.meta
        B, T, Cc = x.shape
        G, H = m["G"], m["H"]
        cpg, ccg = Cc // G, wq.shape[0]
        M = B * T * G
        if _CGA_FUSED_BWD and dout.dtype == torch.bfloat16:
            return CGABranchFn._backward_fused(ctx, dout, x, wq, bq, wk, bk, wv, bv, wproj, bproj, shk, shv, o, sh_k_in, sh_v_in, m, B, T, Cc, G, H, cpg, ccg, M)
        # ---- proj backward: dO = (dout * mask) Wproj ; dWproj, dbproj deferred
        d_o = _linear_bwd(o, wproj, bproj, dout.reshape(B * T, Cc), 0, Cc, True, drop=m["proj_drop"])
        with torch.no_grad():
            # ---- q / k / v again (one GEMM), the attention-core backward, the stacked projection's backward
            x2 = x.reshape(M, cpg)
            pk = pack_for(x.device)
            Wc, Wt = pk.get([wq, wk, wv], x.dtype)
            bias = pk.get([bq, bk, bv], torch.float32)[0].reshape(-1)
            qkv = torch.empty(M, 3 * ccg, dtype=x.dtype, device=x.device)
            K.gemm_nt(x2, Wc, qkv, M, 3 * ccg, cpg, cpg, cpg, 3 * ccg, bias)
            spec = dict(m["spec"])
            dqkv, _, _, _, sk_ret, sv_ret = _attn_bwd(qkv, None, None, None, shk, shv, sh_k_in, sh_v_in, spec, d_o.reshape(M, ccg))
            dx = None
            if ctx.needs_input_grad[0]:
                dx = torch.empty(M, cpg, dtype=x.dtype, device=x.device)
                K.gemm_nt(dqkv, Wt, dx, M, cpg, 3 * ccg, 3 * ccg, 3 * ccg, cpg, None)
            off, esz = 0, dqkv.element_size()
            DeferDW.arm()
            for w, b in ((wq, bq), (wk, bk), (wv, bv)):
                wbuf, _ = grad_sink(w)
                bbuf, _ = grad_sink(b)
                if wbuf is not None:
                    K.gemm_tn(dqkv, x2, wbuf, M, ccg, cpg, 3 * ccg, cpg, cpg, bbuf, A_ptr=dqkv.data_ptr() + off * esz)
                off += ccg
        return (dx.reshape(B, T, Cc) if dx is not None else None), None, None, None, None, None, None, None, None, \
            _ret(sk_ret, sh_k_in) if sk_ret is not None else None, _ret(sv_ret, sh_v_in) if sv_ret is not None else None, None


def _cga_backward_fused(ctx, dout, x, wq, bq, wk, bk, wv, bv, wproj, bproj, shk, shv, o, sh_k_in, sh_v_in, m, B, T, Cc, G, H, cpg, ccg, M):
    """CGABranchFn.backward through csrc/cga.hip's backward kernel: one launch from dout to dx / dqkv / dz; weight gradients deferred
    as everywhere; the bank rows' gradients (sums over images and groups) reduced at once -- BankProj2Fn.backward is waiting for them."""
    rt = _rt(x)
    g2 = dout.reshape(B * T, Cc)
    if not g2.is_contiguous():
        g2 = g2.contiguous()
    pd, ad = m["proj_drop"], m["attn_drop"]
    dz = torch.empty_like(g2) if pd[0] > 0.0 else g2
    pk = pack_for(x.device)
    Wqkv, WqkvT = pk.get([wq, wk, wv], x.dtype)
    bqkv = pk.get([bq, bk, bv], torch.float32)[0].reshape(-1)
    _, WpT = pk.get(wproj, x.dtype)
    dqkv = torch.empty(M, 3 * ccg, dtype=x.dtype, device=x.device)
    dx = torch.empty(B * T, Cc, dtype=x.dtype, device=x.device)
    nparts = int(L.load().qavit_cga_bwd_parts(B, T))
    parts = torch.empty(nparts * 512, dtype=torch.float32, device=x.device)
    a = L.CgaBwdArgs()
    a.dtype = K.dt_code(x.dtype)
    a.B, a.T, a.C, a.G, a.H, a.D, a.S = B, T, Cc, G, H, ccg // H, shk.shape[0]
    a.dout, a.lddout = g2.data_ptr(), Cc
    a.x, a.ldx = x.data_ptr(), Cc
    a.wqkv_rm, a.wqkvT_rm, a.bqkv = Wqkv.data_ptr(), WqkvT.data_ptr(), bqkv.data_ptr()
    a.wprojT_rm = WpT.data_ptr()
    a.sh_k, a.sh_v = shk.data_ptr(), shv.data_ptr()
    a.attn_drop_p, a.attn_drop_site = float(ad[0]), int(ad[1])
    a.proj_drop_p, a.proj_drop_site = float(pd[0]), int(pd[1])
    a.rng = rt.rng.data_ptr()
    if pd[0] > 0.0:
        a.dz, a.lddz = dz.data_ptr(), Cc
    a.dqkv = dqkv.data_ptr()
    a.dx, a.lddx = dx.data_ptr(), Cc
    a.parts = parts.data_ptr()
    if getattr(ctx, "trip", None) is not None:
        a.nan_trip = ctx.trip.data_ptr()
    DeferDW.arm()
    L.check(L.load().qavit_cga_bwd(C.byref(a), K.stream()), "cga_bwd")
    # dW_proj += dz^T O, db_proj += colsum(dz)
    wbuf, _ = grad_sink(wproj)
    bbuf, _ = grad_sink(bproj)
    if wbuf is not None or bbuf is not None:
        if wbuf is None:
            wbuf = torch.zeros(wproj.shape, dtype=torch.float32, device=x.device)
        K.gemm_tn(dz, o, wbuf, B * T, Cc, wproj.shape[1], Cc, wproj.shape[1], wproj.shape[1], bbuf)
    # q / k / v weight gradients from the rows of dqkv and the [B*T*G, 32] view of x
    x2 = x.reshape(M, cpg)
    off, esz = 0, dqkv.element_size()
    for w, b in ((wq, bq), (wk, bk), (wv, bv)):
        wb_, _ = grad_sink(w)
        bb_, _ = grad_sink(b)
        if wb_ is not None:
            K.gemm_tn(dqkv, x2, wb_, M, ccg, cpg, 3 * ccg, cpg, cpg, bb_, A_ptr=dqkv.data_ptr() + off * esz)
        off += ccg
    # bank-row gradients: [d sh_k | d sh_v] partial rows -> the (non-leaf) projected bank rows' gradient buffers
    sk_ret = sv_ret = None
    need_k, need_v = sh_k_in.requires_grad, sh_v_in.requires_grad
    rec = DeferredBank.record_of(sh_k_in, sh_v_in)
    if rec is not None:
        both = DeferredBank.slot(shk.shape, x.device)
        K.DeferredLN.push_raw(parts.data_ptr(), nparts, 256, both[0].data_ptr(), both[1].data_ptr(), 512, (parts, both))
        DeferredBank.queue.append((rec, both))
    elif need_k or need_v:
        both = torch.zeros((2,) + tuple(shk.shape), dtype=torch.float32, device=x.device)
        K.reduce_now([K.DeferredLN.desc(parts.data_ptr(), nparts, 256, both[0].data_ptr() if need_k else None, both[1].data_ptr() if need_v else None, 512)])
        sk_ret = both[0] if need_k else None
        sv_ret = both[1] if need_v else None
    return (dx.reshape(B, T, Cc) if ctx.needs_input_grad[0] else None), None, None, None, None, None, None, None, None, \
        _ret(sk_ret, sh_k_in) if sk_ret is not None else None, _ret(sv_ret, sh_v_in) if sv_ret is not None else None, None


CGABranchFn._backward_fused = staticmethod(_cga_backward_fused)


class AttnFn(Function):
    """See include/qavit.h (qavit_attn_args).  ``q_t`` is a 2-D row matrix holding q (and, when ``kv_t`` is
    None and L > 0, also k and v) at column offsets; gradients come back as whole matrices.
    ``spec["drop"] = (p, site)``: the ``dropout_p`` the reference passes to SDPA (HQAViT_CIFAR100.py:390-392); the
    mask is regenerated in backward from the runtime's (seed, step) words."""

    @staticmethod
    def forward(ctx, q_t, kv_t, E_k, E_v, sh_k, sh_v, spec):
        rt = _rt(q_t)
        s = spec
        HD = s["H"] * s["D"]
        src_kv = q_t if kv_t is None else kv_t
        a = K.attn_args(q_t.dtype, s["mode"], s["G"], s["Nq"], s["L"], s["H"], s["D"], s.get("KC", 0), s["S"],
                        s.get("groups_per_b", 0), s.get("q_rows_per_b", 0), s.get("k_rows_per_b", 0), s.get("q_tbl"), s.get("k_tbl"))
        esz = q_t.element_size()
        o = torch.empty(s["q_rows"], HD, dtype=q_t.dtype, device=q_t.device)
        a.q, a.ldq = q_t.data_ptr() + s["q_off"] * esz, q_t.shape[1]
        if s["L"] > 0:
            a.k_tok, a.ldk = src_kv.data_ptr() + s["k_off"] * esz, src_kv.shape[1]
            a.v_tok, a.ldv = src_kv.data_ptr() + s["v_off"] * esz, src_kv.shape[1]
        if s["mode"] == 0:
            a.E_k, a.E_v = E_k.data_ptr(), E_v.data_ptr()
        a.sh_k, a.sh_v = sh_k.data_ptr(), sh_v.data_ptr()
        a.o, a.ldo = o.data_ptr(), HD
        _attn_drop(a, s, rt)
        guard = rt.nan_guard
        if guard:
            a.nan_flag = rt.nan_flag.data_ptr()
        K.attn_fwd(a)
        if guard:
            K.nan_guard(o, rt.nan_flag)
        ctx.spec = s
        # The bank is mutated in place later in the same forward (GlobalTokenBank.write); the reference's SDPA
        # backward sees the values its forward used (torch.cat made a copy), so snapshot shared rows that alias
        # a parameter.
        ctx.sh_alias = (sh_k, sh_v)
        need = any(ctx.needs_input_grad)
        sk_s, sv_s = sh_k, sh_v
        if need and sh_k.is_leaf and sh_v.is_leaf:
            snap = s.get("bank_snap")
            sk_s, sv_s = snap if snap is not None else K.copy2(sh_k, sh_v)
        elif need and (sh_k.is_leaf or sh_v.is_leaf):
            sk_s = sh_k.detach().clone() if sh_k.is_leaf else sh_k
            sv_s = sh_v.detach().clone() if sh_v.is_leaf else sh_v
        ctx.save_for_backward(q_t, kv_t, E_k, E_v, sk_s, sv_s)
        return o

    @staticmethod
    def backward(ctx, d_o):
        q_t, kv_t, E_k, E_v, sh_k, sh_v = ctx.saved_tensors
        sh_k_in, sh_v_in = ctx.sh_alias
        dq_t, dkv_t, ek_ret, ev_ret, sk_ret, sv_ret = _attn_bwd(q_t, kv_t, E_k, E_v, sh_k, sh_v, sh_k_in, sh_v_in, ctx.spec, d_o)
        return dq_t, dkv_t, _ret(ek_ret, E_k) if ek_ret is not None else None, _ret(ev_ret, E_v) if ev_ret is not None else None, \
            _ret(sk_ret, sh_k_in), _ret(sv_ret, sh_v_in), None


def _attn_bwd(q_t, kv_t, E_k, E_v, sh_k, sh_v, sh_k_in, sh_v_in, s, d_o):
    """Attention-core backward (qavit_attn_bwd): ``sh_k`` / ``sh_v`` are the forward-time VALUES of the shared rows, ``sh_k_in`` /
    ``sh_v_in`` the tensors whose gradient they feed (the bank parameters accumulate in place).  -> dq_t, dkv_t and the
    autograd returns for E_k, E_v, sh_k_in, sh_v_in (None where the gradient went straight into .grad).  Shared by AttnFn and
    BranchFn."""
    rt = _rt(q_t)
    HD = s["H"] * s["D"]
    if not d_o.is_contiguous():
        d_o = d_o.contiguous()
    src_kv = q_t if kv_t is None else kv_t
    a = K.attn_args(q_t.dtype, s["mode"], s["G"], s["Nq"], s["L"], s["H"], s["D"], s.get("KC", 0), s["S"],
                    s.get("groups_per_b", 0), s.get("q_rows_per_b", 0), s.get("k_rows_per_b", 0), s.get("q_tbl"), s.get("k_tbl"))
    esz = q_t.element_size()
    a.q, a.ldq = q_t.data_ptr() + s["q_off"] * esz, q_t.shape[1]
    if s["L"] > 0:
        a.k_tok, a.ldk = src_kv.data_ptr() + s["k_off"] * esz, src_kv.shape[1]
        a.v_tok, a.ldv = src_kv.data_ptr() + s["v_off"] * esz, src_kv.shape[1]
    if s["mode"] == 0:
        a.E_k, a.E_v = E_k.data_ptr(), E_v.data_ptr()
    a.sh_k, a.sh_v = sh_k.data_ptr(), sh_v.data_ptr()
    a.d_o, a.lddo = d_o.data_ptr(), HD
    _attn_drop(a, s, rt)
    covered_q = HD * (3 if (kv_t is None and s["L"] > 0) else 1) == q_t.shape[1]
    dq_t = torch.empty_like(q_t) if covered_q else torch.zeros_like(q_t)
    a.dq, a.lddq = dq_t.data_ptr() + s["q_off"] * esz, q_t.shape[1]
    dkv_t = None
    if s["L"] > 0:
        if kv_t is None:
            dst = dq_t
        else:
            # every row and column written by the kernel?  (MSDA at 224 px keeps only the first 128 landmarks: the rest get 0)
            covered_kv = 2 * HD == kv_t.shape[1] and s["G"] * s["L"] == kv_t.shape[0]
            dkv_t = torch.empty_like(kv_t) if covered_kv else torch.zeros_like(kv_t)
            dst = dkv_t
        a.dk_tok, a.lddk = dst.data_ptr() + s["k_off"] * esz, dst.shape[1]
        a.dv_tok, a.lddv = dst.data_ptr() + s["v_off"] * esz, dst.shape[1]
    ek_buf, ek_ret = grad_sink(E_k) if s["mode"] == 0 else (None, None)
    ev_buf, ev_ret = grad_sink(E_v) if s["mode"] == 0 else (None, None)
    if (sh_k_in is not None and sh_v_in is not None and sh_k_in.requires_grad and sh_v_in.requires_grad and not sh_k_in.is_leaf
            and not sh_v_in.is_leaf and sh_k_in.shape == sh_v_in.shape):
        both = torch.zeros((2,) + tuple(sh_k_in.shape), dtype=torch.float32, device=sh_k_in.device)     # one fill for the two sinks
        sk_buf = sk_ret = both[0]
        sv_buf = sv_ret = both[1]
    else:
        sk_buf, sk_ret = grad_sink(sh_k_in)
        sv_buf, sv_ret = grad_sink(sh_v_in)
    a.dE_k, a.dE_v = K._p(ek_buf), K._p(ev_buf)
    a.dsh_k, a.dsh_v = K._p(sk_buf), K._p(sv_buf)
    nws = K.attn_ws_floats(a)
    ws = rt.workspace("attn_bwd", nws)
    a.ws, a.ws_floats = ws.data_ptr(), ws.numel()
    K.attn_bwd(a)
    return dq_t, dkv_t, ek_ret, ev_ret, sk_ret, sv_ret


class TokMixFn(Function):
    @staticmethod
    def forward(ctx, scores, x):
        B, N, M = scores.shape
        Cc = x.shape[-1]
        scores = scores.contiguous()
        x = x.contiguous()
        p = torch.empty_like(scores)
        xc = torch.empty(B, M, Cc, dtype=x.dtype, device=x.device)
        K.tokmix_fwd(scores, x, p, xc, B, N, M, Cc)
        ctx.save_for_backward(p, x)
        return xc

    @staticmethod
    def backward(ctx, dxc):
        p, x = ctx.saved_tensors
        B, N, M = p.shape
        Cc = x.shape[-1]
        dxc = dxc.contiguous()
        dx = torch.empty_like(x)
        ds = torch.empty_like(p)
        K.tokmix_bwd(p, x, dxc, dx, ds, B, N, M, Cc)
        return ds, dx


_TL_FUSED = os.environ.get("QAVIT_FUSED_TL", "1") != "0"


def tl_ok(x, w) -> bool:
    """TokenLearner as the fused node (TokenLearnerFn)?  bf16 [B, 64, 192] tokens, 16 learned tokens."""
    return bool(_TL_FUSED and x.is_cuda and x.dim() == 3 and w.dim() == 2 and w.shape[1] == x.shape[-1]
                and K.tl_ok(x if x.is_contiguous() else x.contiguous(), x.shape[1], w.shape[0], x.shape[2]))


class TokenLearnerFn(Function):
    """xc = softmax_N(Linear(LayerNorm(x)))^T x  (HQAViT_CIFAR100.py:971-1002) as ONE autograd node, one launch each way (qavit_tl_fwd /
    qavit_tl_bwd).  It replaces LinearFn (LayerNorm-prologue GEMM with 16 outputs) + TokMixFn forward and tokmix_bwd + layernorm_bwd_lin +
    the score Linear's deferred weight-gradient GEMM backward: the [B*N, C] token matrix is read once per direction, and the parameter
    gradients (score weight / bias, LayerNorm gamma / beta) leave the backward kernel as partial rows for the pass's reduce launch."""

    @staticmethod
    def forward(ctx, x, ln_g, ln_b, w, b, eps):
        K._require_cuda(x, w)
        B, N, Cc = x.shape
        M = w.shape[0]
        x = x.contiguous()
        Wc, _ = pack_for(x.device).get(w, x.dtype)
        p = torch.empty(B, N, M, dtype=x.dtype, device=x.device)
        xc = torch.empty(B, M, Cc, dtype=x.dtype, device=x.device)
        mean = torch.empty(B * N, dtype=torch.float32, device=x.device)
        rstd = torch.empty(B * N, dtype=torch.float32, device=x.device)
        K.tl_fwd(x, ln_g, ln_b, eps, Wc, None if b is None else b.detach(), p, xc, mean, rstd, B, N, M, Cc)
        ctx.save_for_backward(x, ln_g, ln_b, w, b, p, mean, rstd)
        return xc

    @staticmethod
    def backward(ctx, dxc):
        x, ln_g, ln_b, w, b, p, mean, rstd = ctx.saved_tensors
        B, N, Cc = x.shape
        M = w.shape[0]
        dxc = dxc.contiguous()
        Wc, _ = pack_for(x.device).get(w, x.dtype)
        dx = torch.empty_like(x)
        wbuf, wret = grad_sink(w)
        bbuf, bret = grad_sink(b)
        gbuf, gret = grad_sink(ln_g)
        bebuf, beret = grad_sink(ln_b)
        DeferDW.arm()
        K.tl_bwd(dxc, x, p, mean, rstd, ln_g, ln_b, Wc, dx, wbuf, bbuf, gbuf, bebuf, B, N, M, Cc)
        return dx, _ret(gret, ln_g), _ret(beret, ln_b), _ret(wret, w), (None if b is None else _ret(bret, b)), None


class UpMixFn(Function):
    @staticmethod
    def forward(ctx, xc, W, bias, g, b, eps):
        B, M, Cc = xc.shape
        N = W.shape[0]
        xc = xc.contiguous()
        y = torch.empty(B, N, Cc, dtype=xc.dtype, device=xc.device)
        mean = torch.empty(B * N, dtype=torch.float32, device=xc.device)
        rstd = torch.empty(B * N, dtype=torch.float32, device=xc.device)
        K.upmix_fwd(xc, W.detach(), bias.detach(), g.detach(), b.detach(), eps, y, mean, rstd, B, N, M, Cc)
        ctx.save_for_backward(xc, W, bias, g, b, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, W, bias, g, b, mean, rstd = ctx.saved_tensors
        B, M, Cc = xc.shape
        N = W.shape[0]
        dy = dy.contiguous()
        dxc = torch.empty_like(xc)
        wbuf, _ = grad_sink(W)
        bbuf, _ = grad_sink(bias)
        gbuf, _ = grad_sink(g)
        bebuf, _ = grad_sink(b)
        if wbuf is None:
            wbuf = torch.zeros_like(W, dtype=torch.float32)
        if gbuf is None:
            gbuf = torch.zeros_like(g, dtype=torch.float32)
        if bebuf is None:
            bebuf = torch.zeros_like(b, dtype=torch.float32)
        K.upmix_bwd(dy, xc, W.detach(), bias.detach(), g.detach(), mean, rstd, dxc, wbuf, bbuf, gbuf, bebuf, B, N, M, Cc)
        return dxc, None, None, None, None, None


class UpMixScaleAddFn(Function):
    """TokenUpMix applied to the block tail: y = upmix(x + droppath(gamma * u)) (HQAViT_CIFAR100.py:1085, :1118-1121) as one autograd node.
    Forward = ONE launch (qavit_upmix_fwd_sa forms xc while it stages the image; else the scale-add launch + the up-mix launch); backward = ONE launch: the up-mix backward also writes du and dgamma
    (qavit_upmix_bwd_sa), where the separate scale-add backward re-read dxc and u from memory once per block."""

    @staticmethod
    def forward(ctx, x, u, gamma, dp, W, bias, g, b, eps):
        rt = _rt(x)
        B, M, Cc = x.shape
        N = W.shape[0]
        x = x.contiguous()
        u = u.contiguous()
        xc = torch.empty_like(x)
        y = torch.empty(B, N, Cc, dtype=xc.dtype, device=xc.device)
        mean = torch.empty(B * N, dtype=torch.float32, device=xc.device)
        rstd = torch.empty(B * N, dtype=torch.float32, device=xc.device)
        gm = None if gamma is None else gamma.detach()
        if _UPMIX_FWD_SA and dp[2] == M and K.upmix_fwd_sa_ok(x, u, N, M, Cc):
            K.upmix_fwd_sa(x, u, gm, dp, rt.rng, xc, W.detach(), bias.detach(), g.detach(), b.detach(), eps, y, mean, rstd, B, N, M, Cc)
        else:
            K.scale_add_fwd(x, u, gm, xc, B * M, Cc, dp, rt.rng)
            K.upmix_fwd(xc, W.detach(), bias.detach(), g.detach(), b.detach(), eps, y, mean, rstd, B, N, M, Cc)
        ctx.save_for_backward(xc, u, gamma, W, bias, g, b, mean, rstd)
        ctx.dp = dp
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, u, gamma, W, bias, g, b, mean, rstd = ctx.saved_tensors
        rt = _rt(xc)
        B, M, Cc = xc.shape
        N = W.shape[0]
        dy = dy.contiguous()
        dxc = torch.empty_like(xc)
        du = torch.empty_like(u)
        wbuf, _ = grad_sink(W)
        bbuf, _ = grad_sink(bias)
        gbuf, _ = grad_sink(g)
        bebuf, _ = grad_sink(b)
        if wbuf is None:
            wbuf = torch.zeros_like(W, dtype=torch.float32)
        if gbuf is None:
            gbuf = torch.zeros_like(g, dtype=torch.float32)
        if bebuf is None:
            bebuf = torch.zeros_like(b, dtype=torch.float32)
        sgbuf, sgret = grad_sink(gamma)
        if K.upmix_bwd_sa_ok(xc, N, M, Cc) and u.data_ptr() % 8 == 0:
            K.upmix_bwd(dy, xc, W.detach(), bias.detach(), g.detach(), mean, rstd, dxc, wbuf, bbuf, gbuf, bebuf, B, N, M, Cc,
                        sa=(u, du, None if gamma is None else gamma.detach(), sgbuf, ctx.dp, rt.rng))
        else:
            K.upmix_bwd(dy, xc, W.detach(), bias.detach(), g.detach(), mean, rstd, dxc, wbuf, bbuf, gbuf, bebuf, B, N, M, Cc)
            K.scale_add_bwd(dxc, u, None if gamma is None else gamma.detach(), du, sgbuf, B * M, Cc, ctx.dp, rt.rng)
        return dxc, du, sgret, None, None, None, None, None, None


class GatherPoolFn(Function):
    @staticmethod
    def forward(ctx, x, idx, stride):
        B, N, Cc = x.shape
        NP = idx.numel() // stride
        x = x.contiguous()
        y = torch.empty(B, NP, Cc, dtype=x.dtype, device=x.device)
        K.gather_pool_fwd(x, idx, y, B, N, NP, stride, Cc)
        ctx.idx, ctx.dims = idx, (B, N, NP, stride, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, NP, stride, Cc = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty(B, N, Cc, dtype=dy.dtype, device=dy.device)
        K.gather_pool_bwd(dy, ctx.idx, dx, B, N, NP, stride, Cc)
        return dx, None, None


class TokenMeanFn(Function):
    @staticmethod
    def forward(ctx, x):
        B, N, Cc = x.shape
        x = x.contiguous()
        y = torch.empty(B, Cc, dtype=x.dtype, device=x.device)
        K.token_mean_fwd(x, y, B, N, Cc)
        ctx.dims = (B, N, Cc)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, N, Cc = ctx.dims
        dy = dy.contiguous()
        dx = torch.empty(B, N, Cc, dtype=dy.dtype, device=dy.device)
        K.token_mean_bwd(dy, dx, B, N, Cc)
        return dx


# ---------------------------------------------------------------------------------------------------
# CCF-FFN middle
# ---------------------------------------------------------------------------------------------------
class CcfMidFn(Function):
    @staticmethod
    def forward(ctx, h, g1, b1, g2, b2, w, cbias, cscale, Hs, Ws, eps):
        B, N, Cc = h.shape
        h = h.contiguous()
        flags = (1 if g1 is not None else 0) | (2 if cbias is not None else 0) | (4 if cscale is not None else 0)
        a = K.ccf_args(h.dtype, flags, B, Hs, Ws, Cc, eps)
        out = torch.empty_like(h)
        stats = [torch.empty(B * N, dtype=torch.float32, device=h.device) for _ in range(4)] if flags & 1 else [None] * 4
        a.h, a.out = h.data_ptr(), out.data_ptr()
        a.g1, a.b1, a.g2, a.b2 = K._p(g1), K._p(b1), K._p(g2), K._p(b2)
        a.w, a.cbias, a.cscale = w.data_ptr(), K._p(cbias), K._p(cscale)
        a.mean1, a.rstd1, a.mean2, a.rstd2 = [K._p(t) for t in stats]
        K.ccf_fwd(a)
        ctx.dims = (B, Hs, Ws, Cc, eps, flags)
        ctx.save_for_backward(h, g1, b1, g2, b2, w, cbias, cscale, *stats)
        return out

    @staticmethod
    def backward(ctx, d_out):
        h, g1, b1, g2, b2, w, cbias, cscale, m1, r1, m2, r2 = ctx.saved_tensors
        B, Hs, Ws, Cc, eps, flags = ctx.dims
        d_out = d_out.contiguous()
        a = K.ccf_args(h.dtype, flags, B, Hs, Ws, Cc, eps)
        d_h = torch.empty_like(h)
        a.h, a.d_out, a.d_h = h.data_ptr(), d_out.data_ptr(), d_h.data_ptr()
        a.g1, a.b1, a.g2, a.b2 = K._p(g1), K._p(b1), K._p(g2), K._p(b2)
        a.w, a.cbias, a.cscale = w.data_ptr(), K._p(cbias), K._p(cscale)
        a.mean1, a.rstd1, a.mean2, a.rstd2 = K._p(m1), K._p(r1), K._p(m2), K._p(r2)
        sinks = [grad_sink(t)[0] for t in (g1, b1, g2, b2, w, cbias, cscale)]
        a.dg1, a.db1, a.dg2, a.db2, a.dw, a.dcbias, a.dcscale = [K._p(t) for t in sinks]
        if a.dw is None:
            scratch = torch.zeros_like(w, dtype=torch.float32)
            a.dw = scratch.data_ptr()
        parts = None
        DeferDW.arm()
        if K.DeferredLN.enabled and K.DeferredLN.ON and Cc <= 256 and Cc % 8 == 0 and Hs * Ws >= 15:
            # per-workgroup partial rows [dg1 | db1 | dg2 | db2 | dcbias | dcscale | dw(9C)], folded when backward ends
            npart = int(L.load().qavit_ccf_bwd_parts(B))
            parts = torch.empty(npart * 15 * Cc, dtype=torch.float32, device=h.device)
            a.parts = parts.data_ptr()
        K.ccf_bwd(a)
        if parts is not None:
            base, st = parts.data_ptr(), 15 * Cc
            sg1, sb1, sg2, sb2, sw, scb, scs = sinks
            keep = (parts,) + tuple(sinks)
            if flags & 1:
                K.DeferredLN.push_raw(base, npart, Cc, K._p(sg1), K._p(sb1), st, keep)
                K.DeferredLN.push_raw(base + 2 * Cc * 4, npart, Cc, K._p(sg2), K._p(sb2), st, keep)
            if scb is not None or scs is not None:
                K.DeferredLN.push_raw(base + 4 * Cc * 4, npart, Cc, K._p(scb), K._p(scs), st, keep)
            if sw is not None:
                half = (9 * Cc) // 2
                K.DeferredLN.push_raw(base + 6 * Cc * 4, npart, half, sw.data_ptr(), sw.data_ptr() + half * 4, st, keep)
        return (d_h,) + (None,) * 10


class Im2ColFn(Function):
    """Rows (b,oy,ox) x columns (c,dy,dx) of a k x k / stride / pad convolution.  ``src`` is the fp32 NCHW image
    (no gradient) or channel-last tokens [B,H*W,Cin] (gradient by col2im)."""

    @staticmethod
    def forward(ctx, src, dims, dtype):
        B, Cin, H, W, k, stride, pad = dims
        nchw = src.dim() == 4
        src = src.contiguous()
        if nchw:
            src = src.float()
        Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
        # rows padded to a multiple of 8 elements (zeros): K = 27 of the first stem convolution as 64-byte rows -- its weight-gradient GEMM over
        # 262144 rows took the generic kernel at 0.33 TB/s on 54-byte rows.  The consumer is linear(..., xpad=True).
        cols = torch.empty(B * Ho * Wo, (Cin * k * k + 7) // 8 * 8, dtype=dtype, device=src.device)
        K.im2col(src, nchw, cols, B, Cin, H, W, k, stride, pad)
        ctx.dims, ctx.nchw = dims, nchw
        return cols

    @staticmethod
    def backward(ctx, dcols):
        if ctx.nchw:
            return None, None, None
        B, Cin, H, W, k, stride, pad = ctx.dims
        if dcols.shape[1] != Cin * k * k:
            raise NotImplementedError("col2im of padded im2col rows (Cin*k*k not a multiple of 8 with a differentiable source)")
        dcols = dcols.contiguous()
        dx = torch.empty(B, H * W, Cin, dtype=dcols.dtype, device=dcols.device)
        K.col2im(dcols, dx, B, Cin, H, W, k, stride, pad)
        return dx, None, None


class SpatialLayerNormFn(Function):
    """nn.LayerNorm([C,H,W]) on channel-last tokens [B, H*W, C]; weight / bias are the module's own [C,H,W] parameters
    (csrc/spatial_ln.hip; HQAViTv2_CIFAR100.py:766)."""

    @staticmethod
    def forward(ctx, x, w, b, eps):
        B, N, Cc = x.shape
        x = x.contiguous()
        y = torch.empty_like(x)
        mean = torch.empty(B, dtype=torch.float32, device=x.device)
        rstd = torch.empty(B, dtype=torch.float32, device=x.device)
        K.spatial_ln_fwd(x, w.detach(), b.detach(), y, mean, rstd, B, N, Cc, eps)
        ctx.save_for_backward(x, w, b, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, b, mean, rstd = ctx.saved_tensors
        B, N, Cc = x.shape
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        wbuf, wret = grad_sink(w)
        bbuf, bret = grad_sink(b)
        if wbuf is None:
            wbuf = torch.zeros_like(w, dtype=torch.float32)
        if bbuf is None:
            bbuf = torch.zeros_like(b, dtype=torch.float32)
        K.spatial_ln_bwd(dy, x, w.detach(), mean, rstd, dx, wbuf, bbuf, B, N, Cc)
        return dx, _ret(wret, w), _ret(bret, b), None


def ccf_mid(h, g1, b1, g2, b2, w, cbias, cscale, Hs, Ws, eps=1e-5):
    """LN -> depthwise 3x3 (+bias) * scale -> LN on [B, Hs*Ws, C] tokens.  One kernel each way while the image tile
    fits LDS (32 px and 64 px models); larger maps (14x14x96 at 224 px) compose the same math from the LayerNorm and
    depthwise-conv kernels, with the per-channel scale folded into the taps: (conv(a; w) + b) * s = conv(a; w*s) + b*s."""
    B, N, Cc = h.shape
    if (4 * N * Cc + 15 * Cc) * 4 <= 160 * 1024:
        return CcfMidFn.apply(h, g1, b1, g2, b2, w, cbias, cscale, Hs, Ws, eps)
    a = layer_norm(h, g1, b1, eps) if g1 is not None else h
    if cscale is not None:
        sc = cscale.reshape(-1)
        w = w * sc.reshape(-1, 1, 1, 1)
        cbias = cbias * sc if cbias is not None else None
    t = DwConvFn.apply(a, w, cbias, Hs, Ws)
    return layer_norm(t, g2, b2, eps) if g2 is not None else t


class DwConvFn(Function):
    """Depthwise k x k conv on channel-last tokens [B, H*W, C] (csrc/dwconv.hip).  ``alias=True`` -> (y, x_alias): use ``x_alias`` for
    the residual connection around the block this convolution opens (ConvNeXtBlock); the gradient arriving on it is added inside
    the backward kernel (qavit_dwconv_bwd_ld's addend) instead of by an elementwise add of autograd's."""

    @staticmethod
    def forward(ctx, x, w, bias, H, W, alias=False):
        B, N, Cc = x.shape
        ks = w.shape[-1]
        x = x.contiguous()
        y = torch.empty_like(x)
        K.dwconv_fwd(x, w.detach(), None if bias is None else bias.detach(), y, B, H, W, Cc, ks)
        ctx.save_for_backward(x, w, bias)
        ctx.dims = (B, H, W, Cc, ks)
        if alias:
            ctx.set_materialize_grads(False)
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dalias=None):
        x, w, bias = ctx.saved_tensors
        B, H, W, Cc, ks = ctx.dims
        if dy is None:                                      # only the alias was differentiated through
            return dalias, None, None, None, None, None
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        wbuf, wret = grad_sink(w)
        bbuf, bret = grad_sink(bias)
        if wbuf is None:
            wbuf = torch.zeros_like(w, dtype=torch.float32)
        fold = dalias is not None and H % 8 == 0 and W % 8 == 0 and dalias.dtype == dy.dtype
        if fold:
            K.dwconv_bwd_ld(dy, Cc, x, w.detach(), dx, dalias.contiguous(), Cc, wbuf, bbuf, B, H, W, Cc, ks)
        else:
            K.dwconv_bwd(dy, x, w.detach(), dx, wbuf, bbuf, B, H, W, Cc, ks)
            if dalias is not None:
                dx = dx + dalias
        return dx, _ret(wret, w), _ret(bret, bias), None, None, None


class LmfGatherFn(Function):
    """LMFAdapter's  cat([dwconv_3x3(x), dwconv_5x5(x), x], channel)  (HQAViT_CIFAR100.py:830-834) on channel-last tokens as one autograd
    node: the two depthwise convolutions write their column slices of the [B, N, 3C] buffer themselves, and backward reads the gradient's
    slices in place -- dx = dw3^T(d0) + dw5^T(d1) + d2 leaves the second convolution's kernel complete (no cat, no slice copies, no adds)."""

    @staticmethod
    def forward(ctx, x, w3, b3, w5, b5, H, W):
        B, N, Cc = x.shape
        x = x.contiguous()
        cat = torch.empty(B, N, 3 * Cc, dtype=x.dtype, device=x.device)
        K.dwconv_fwd_ld(x, w3.detach(), None if b3 is None else b3.detach(), cat, 3 * Cc, B, H, W, Cc, w3.shape[-1])
        if H % 8 == 0 and W % 8 == 0:                       # the pass-through member x rides in the second convolution's launch
            K.dwconv_fwd_ld2(x, w5.detach(), None if b5 is None else b5.detach(), cat[:, :, Cc:], 3 * Cc, cat[:, :, 2 * Cc:], 3 * Cc, B, H, W, Cc, w5.shape[-1])
        else:
            K.dwconv_fwd_ld(x, w5.detach(), None if b5 is None else b5.detach(), cat[:, :, Cc:], 3 * Cc, B, H, W, Cc, w5.shape[-1])
            cat[:, :, 2 * Cc:].copy_(x)
        ctx.save_for_backward(x, w3, b3, w5, b5)
        ctx.dims = (B, H, W, Cc)
        return cat

    @staticmethod
    def backward(ctx, dcat):
        x, w3, b3, w5, b5 = ctx.saved_tensors
        B, H, W, Cc = ctx.dims
        dcat = dcat.contiguous()
        dx = torch.empty_like(x)
        sinks = []
        for p, needed in ((w3, True), (b3, False), (w5, True), (b5, False)):
            buf, ret = grad_sink(p)
            if buf is None and needed:                       # the kernel always accumulates a weight gradient
                buf = torch.zeros_like(p, dtype=torch.float32)
            sinks.append((buf, ret))
        ld = 3 * Cc
        K.dwconv_bwd_ld(dcat, ld, x, w3.detach(), dx, dcat[:, :, 2 * Cc:], ld, sinks[0][0], sinks[1][0], B, H, W, Cc, w3.shape[-1])
        K.dwconv_bwd_ld(dcat[:, :, Cc:], ld, x, w5.detach(), dx, dx, Cc, sinks[2][0], sinks[3][0], B, H, W, Cc, w5.shape[-1])
        return dx, _ret(sinks[0][1], w3), _ret(sinks[1][1], b3), _ret(sinks[2][1], w5), _ret(sinks[3][1], b5), None, None


# ---------------------------------------------------------------------------------------------------
# elementwise helpers
# ---------------------------------------------------------------------------------------------------
class HybridFuseFn(Function):
    @staticmethod
    def forward(ctx, x, fw):
        nb = fw.numel()
        Cc = x.shape[-1]
        x = x.contiguous()
        rows = x.numel() // Cc
        y = torch.empty_like(x)
        K.hybrid_fuse_fwd(x, fw.detach(), y, rows, nb, Cc // nb)
        ctx.save_for_backward(x, fw)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, fw = ctx.saved_tensors
        nb = fw.numel()
        Cc = x.shape[-1]
        rows = x.numel() // Cc
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        fbuf, fret = grad_sink(fw)
        if fbuf is None:
            fbuf = torch.zeros(nb, dtype=torch.float32, device=x.device)
        K.hybrid_fuse_bwd(dy, x, fw.detach(), dx, fbuf, rows, nb, Cc // nb)
        return dx, fret


class CompressFuseFn(Function):
    """QuadAttentionBlock's four  compress_i(norm_i(branch_i))  -> concat -> HybridFusion  (HQAViT_CIFAR100.py:904-925)
    as one autograd node: each Linear(C -> C/4) with its LayerNorm prologue writes straight into its column slice of the
    concatenated buffer (ldc = 4*C/4), and in backward reads its slice of the hybrid-fuse gradient in place (lda) --
    no torch.cat, no .contiguous() copies of the four gradient slices.

    apply(fw, eps, x_0, g_0, b_0, W_0, bias_0, ..., x_3, g_3, b_3, W_3, bias_3)"""

    @staticmethod
    def forward(ctx, fw, eps, *args):
        nb = len(args) // 5
        xs = [args[5 * i].reshape(-1, args[5 * i].shape[-1]) for i in range(nb)]
        xs = [x if x.is_contiguous() else x.contiguous() for x in xs]
        K._require_cuda(xs[0], fw)
        rt = _rt(xs[0])
        M, Kd = xs[0].shape
        dev, dt = xs[0].device, xs[0].dtype
        Cb = args[3].shape[0]
        cat = torch.empty(M, nb * Cb, dtype=dt, device=dev)
        esz = cat.element_size()
        means = [torch.empty(M, dtype=torch.float32, device=dev) for _ in range(nb)]
        rstds = [torch.empty(M, dtype=torch.float32, device=dev) for _ in range(nb)]
        T_ = args[0].shape[-2] if args[0].dim() >= 2 else 0
        if T_ > 16 and T_ % 16 == 0:
            T_ = 16                                         # the node is token-wise: 64 tokens per image = four 16-token problems to the kernel
        fix = None
        if rt.pending_fix is not None:                      # the branch that produced one of the operands left its NaN rule to this launch
            fix, fout, _keep = rt.pending_fix
            rt.pending_fix = None
            which = [i for i in range(nb) if xs[i].data_ptr() == fout.data_ptr()]
            if not which:
                raise RuntimeError("the deferred NaN rule belongs to a tensor that is not an operand of this compress-fuse node")
            fix_branch = which[0]
        if (_CFUSE and dt == torch.bfloat16 and nb == 4 and M % max(T_, 1) == 0 and all(a_ is not None for a_ in args[:20])
                and L.load().qavit_compress_fuse_supported(T_, Kd, nb, Cb)):
            # norms, compress Linears, concat and fusion scaling in ONE launch (csrc/cfuse.hip)
            a = L.CfuseArgs()
            a.dtype, a.B, a.T, a.C, a.NB, a.CB = K.dt_code(dt), M // T_, T_, Kd, nb, Cb
            pk = pack_for(dev)
            for i in range(nb):
                g, b, W, bias = args[5 * i + 1: 5 * i + 5]
                a.x[i], a.gamma[i], a.beta[i] = xs[i].data_ptr(), g.data_ptr(), b.data_ptr()
                a.w_rm[i] = pk.get(W, dt)[0].data_ptr()
                a.bias[i] = None if bias is None else bias.data_ptr()
                a.mean[i], a.rstd[i] = means[i].data_ptr(), rstds[i].data_ptr()
            y = torch.empty_like(cat)
            a.fw, a.eps, a.cat, a.y = fw.data_ptr(), float(eps), cat.data_ptr(), y.data_ptr()
            if fix is not None:
                a.fix, a.fix_branch = fix, fix_branch
            L.check(L.load().qavit_compress_fuse_fwd(C.byref(a), K.stream()), "compress_fuse_fwd")
            stats = [t for i in range(nb) for t in (means[i], rstds[i])]
            ctx.meta = (nb, M, Kd, Cb, eps, args[0].shape)
            ctx.save_for_backward(fw, cat, *xs, *[a_ for i in range(nb) for a_ in args[5 * i + 1: 5 * i + 5]], *stats)
            return y.reshape(*args[0].shape[:-1], nb * Cb)
        if fix is not None:                                  # this route cannot carry the rule: its own launch first
            L.check(L.load().qavit_branch_nan_fix(K.dt_code(dt), xs[fix_branch].data_ptr(), M, Kd, C.byref(fix), K.stream()), "branch_nan_fix")
        K.row_stats_multi(xs, eps, M, Kd, means, rstds)                      # the four branch norms: one grid
        probs, stats = [], []
        for i in range(nb):
            g, b, W, bias = args[5 * i + 1: 5 * i + 5]
            Wc, _ = pack_for(dev).get(W, dt)
            probs.append(K.gemm_nt(xs[i], Wc, cat, M, Cb, Kd, Kd, Kd, nb * Cb, None if bias is None else bias.detach(), a_mode=1,
                                   ln=(g, b, eps), ln_stats=(means[i], rstds[i]), rng=rt.rng, C_ptr=cat.data_ptr() + i * Cb * esz,
                                   build_only=True))
            stats += [means[i], rstds[i]]
        K.gemm_nt_grouped(probs)                                             # the four compress Linears: one grid
        y = torch.empty_like(cat)
        K.hybrid_fuse_fwd(cat, fw.detach(), y, M, nb, Cb)
        ctx.meta = (nb, M, Kd, Cb, eps, args[0].shape)
        ctx.save_for_backward(fw, cat, *xs, *[a for i in range(nb) for a in args[5 * i + 1: 5 * i + 5]], *stats)
        return y.reshape(*args[0].shape[:-1], nb * Cb)

    @staticmethod
    def backward(ctx, dy):
        nb, M, Kd, Cb, eps, xshape = ctx.meta
        sv = ctx.saved_tensors
        fw, cat = sv[0], sv[1]
        xs = sv[2:2 + nb]
        prm = sv[2 + nb:2 + nb + 4 * nb]
        stats = sv[2 + 5 * nb:]
        dev, dt = cat.device, cat.dtype
        rt = _rt(cat)
        dy = dy.reshape(M, nb * Cb)
        if not dy.is_contiguous():
            dy = dy.contiguous()
        dcat = torch.empty_like(cat)
        fbuf, fret = grad_sink(fw)
        if fbuf is None:
            fbuf = torch.zeros(nb, dtype=torch.float32, device=dev)
        T_ = xshape[-2] if len(xshape) >= 2 else 0
        if T_ > 16 and T_ % 16 == 0:
            T_ = 16
        gs_all = [grad_sink(prm[4 * i]) for i in range(nb)]
        bs_all = [grad_sink(prm[4 * i + 1]) for i in range(nb)]
        if (_CFUSE_BWD and dt == torch.bfloat16 and nb == 4 and T_ > 0 and all(g_[0] is not None for g_ in gs_all) and all(b_[0] is not None for b_ in bs_all)
                and K.DeferredLN.ON and L.load().qavit_compress_fuse_supported(T_, Kd, nb, Cb)):
            # scaling backward, the four input-gradient GEMMs and the four LayerNorm backwards in ONE launch (csrc/cfuse.hip)
            DeferDW.arm()
            a = L.CfuseBwdArgs()
            a.dtype, a.B, a.T, a.C, a.NB, a.CB = K.dt_code(dt), M // T_, T_, Kd, nb, Cb
            pk = pack_for(dev)
            dxs = [torch.empty(M, Kd, dtype=dt, device=dev) for _ in range(nb)]
            for i in range(nb):
                a.x[i], a.gamma[i] = xs[i].data_ptr(), prm[4 * i].data_ptr()
                a.w_rm[i] = pk.get(prm[4 * i + 2], dt)[0].data_ptr()
                a.mean[i], a.rstd[i] = stats[2 * i].data_ptr(), stats[2 * i + 1].data_ptr()
                a.dx[i] = dxs[i].data_ptr()
            npart = int(L.load().qavit_compress_fuse_bwd_parts(M // T_))
            PF = 1544
            parts = torch.empty(npart * PF, dtype=torch.float32, device=dev)
            a.dy, a.cat, a.fw, a.dcat, a.parts = dy.data_ptr(), cat.data_ptr(), fw.data_ptr(), dcat.data_ptr(), parts.data_ptr()
            L.check(L.load().qavit_compress_fuse_bwd(C.byref(a), K.stream()), "compress_fuse_bwd")
            keep = (parts, fbuf) + tuple(g_[0] for g_ in gs_all) + tuple(b_[0] for b_ in bs_all)
            for i in range(nb):
                K.DeferredLN.push_raw(parts.data_ptr() + i * 2 * Kd * 4, npart, Kd, gs_all[i][0].data_ptr(), bs_all[i][0].data_ptr(), PF, keep)
            K.DeferredLN.push_raw(parts.data_ptr() + nb * 2 * Kd * 4, npart, 4, fbuf.data_ptr(), None, PF, keep)
            esz = dcat.element_size()
            grads = []
            for i in range(nb):
                g, b, W, bias = prm[4 * i: 4 * i + 4]
                mean, rstd = stats[2 * i], stats[2 * i + 1]
                wbuf, wret = grad_sink(W)
                bbuf2, b2ret = grad_sink(bias)
                if wbuf is None:
                    wbuf = torch.zeros(W.shape, dtype=torch.float32, device=dev)
                K.gemm_tn(dcat, xs[i], wbuf, M, Cb, Kd, nb * Cb, Kd, Kd, bbuf2, ln=(g, b, mean, rstd), A_ptr=dcat.data_ptr() + i * Cb * esz)
                grads += [dxs[i].reshape(xshape), _ret(gs_all[i][1], g), _ret(bs_all[i][1], b), _ret(wret, W), None if bias is None else _ret(b2ret, bias)]
            return (fret, None, *grads)
        K.hybrid_fuse_bwd(dy, cat, fw.detach(), dcat, fbuf, M, nb, Cb)
        esz = dcat.element_size()
        DeferDW.arm()
        dxns = [torch.empty(M, Kd, dtype=dt, device=dev) for _ in range(nb)]
        probs = []
        for i in range(nb):
            _, Wt = pack_for(dev).get(prm[4 * i + 2], dt)
            probs.append(K.gemm_nt(dcat, Wt, dxns[i], M, Kd, Cb, nb * Cb, Wt.shape[1], Kd, None, rng=rt.rng,
                                   A_ptr=dcat.data_ptr() + i * Cb * esz, build_only=True))
        K.gemm_nt_grouped(probs)                                             # four input-gradient GEMMs: one grid
        gsinks = [grad_sink(prm[4 * i]) for i in range(nb)]
        bsinks = [grad_sink(prm[4 * i + 1]) for i in range(nb)]
        dxs = [torch.empty_like(dxns[i]) for i in range(nb)]
        if all(gs[0] is not None for gs in gsinks) and all(bs[0] is not None for bs in bsinks):
            K.layernorm_bwd_multi(dxns, xs, [prm[4 * i] for i in range(nb)], [stats[2 * i] for i in range(nb)], [stats[2 * i + 1] for i in range(nb)],
                                  dxs, [gs[0] for gs in gsinks], [bs[0] for bs in bsinks], M, Kd)    # four LayerNorm backwards: one grid
        else:
            for i in range(nb):
                K.layernorm_bwd(dxns[i], xs[i], prm[4 * i], stats[2 * i], stats[2 * i + 1], dxs[i], gsinks[i][0], bsinks[i][0], M, Kd)
        grads = []
        for i in range(nb):
            g, b, W, bias = prm[4 * i: 4 * i + 4]
            mean, rstd = stats[2 * i], stats[2 * i + 1]
            a_ptr = dcat.data_ptr() + i * Cb * esz
            wbuf, wret = grad_sink(W)
            bbuf2, b2ret = grad_sink(bias)
            if wbuf is None:
                wbuf = torch.zeros(W.shape, dtype=torch.float32, device=dev)
            K.gemm_tn(dcat, xs[i], wbuf, M, Cb, Kd, nb * Cb, Kd, Kd, bbuf2, ln=(g, b, mean, rstd), A_ptr=a_ptr)
            grads += [dxs[i].reshape(xshape), _ret(gsinks[i][1], g), _ret(bsinks[i][1], b), _ret(wret, W), None if bias is None else _ret(b2ret, bias)]
        return (fret, None, *grads)


_MLP2 = os.environ.get("QAVIT_FUSED_MLP", "1") != "0"


def mlp2_ok(y, resid, w1, w2) -> bool:
    """Does csrc/mlp2.hip cover this bottleneck MLP?  (bf16 rows of 192 channels, hidden 96)"""
    return (_MLP2 and y.is_cuda and y.dtype == torch.bfloat16 and resid.dtype == torch.bfloat16 and y.shape == resid.shape and w1.dim() == 2 and
            tuple(w2.shape) == (w1.shape[1], w1.shape[0]) and bool(L.load().qavit_mlp2_supported(w1.shape[1], w1.shape[0])) and y.shape[-1] == w1.shape[1])


class Mlp2Fn(Function):
    """x1 = resid + drop_path(dropout(fc2(dropout(GELU(fc1(y)))))) -- BottleneckMLP + residual (HQAViT_CIFAR100.py:643-656, :1082-1083) --
    as ONE launch forward and ONE backward (csrc/mlp2.hip) instead of two GEMM launches each way.  ``opts``: drop1 / drop2 = (p, site),
    dp = (p, site, rows per sample).  The residual's gradient is the incoming gradient itself (returned as is: the caller's alias mechanics
    add it where the residual came from); weight gradients are deferred grouped GEMMs on the operands the kernels write."""

    @staticmethod
    def forward(ctx, y, resid, w1, b1, w2, b2, opts):
        K._require_cuda(y, w1)
        rt = _rt(y)
        C_ = y.shape[-1]
        Hd = w1.shape[0]
        y2 = y.reshape(-1, C_)
        r2 = resid.reshape(-1, C_)
        y2 = y2 if y2.is_contiguous() else y2.contiguous()
        r2 = r2 if r2.is_contiguous() else r2.contiguous()
        M = y2.shape[0]
        pk = pack_for(y.device)
        W1c, W2c = pk.get(w1, y.dtype)[0], pk.get(w2, y.dtype)[0]
        need = any(ctx.needs_input_grad)
        out = torch.empty(M, C_, dtype=y.dtype, device=y.device)
        z1 = torch.empty(M, Hd, dtype=y.dtype, device=y.device) if need else None
        h1 = torch.empty(M, Hd, dtype=y.dtype, device=y.device) if need else None
        d1, d2, dp = opts.get("drop1") or (0.0, 0), opts.get("drop2") or (0.0, 0), opts.get("dp") or (0.0, 0, 1)
        a = L.Mlp2Args()
        a.dtype, a.M, a.C, a.Hd = K.dt_code(y.dtype), M, C_, Hd
        a.y, a.ldy, a.resid, a.ldr = y2.data_ptr(), C_, r2.data_ptr(), C_
        a.w1_rm, a.b1, a.w2_rm, a.b2 = W1c.data_ptr(), b1.data_ptr(), W2c.data_ptr(), b2.data_ptr()
        a.drop1_p, a.drop1_site, a.drop2_p, a.drop2_site = float(d1[0]), int(d1[1]), float(d2[0]), int(d2[1])
        a.dp_p, a.dp_site, a.dp_rows = float(dp[0]), int(dp[1]), int(dp[2])
        a.rng = rt.rng.data_ptr()
        a.out, a.ldo = out.data_ptr(), C_
        a.z1, a.h1 = K._p(z1), K._p(h1)
        L.check(L.load().qavit_mlp2_fwd(C.byref(a), K.stream()), "mlp2_fwd")
        if need:
            ctx.meta = (M, C_, Hd, d1, d2, dp, y.shape)
            ctx.save_for_backward(y2, w1, b1, w2, b2, z1, h1)
        return out.reshape(y.shape)

    @staticmethod
    def backward(ctx, g):
        y2, w1, b1, w2, b2, z1, h1 = ctx.saved_tensors
        M, C_, Hd, d1, d2, dp, yshape = ctx.meta
        rt = _rt(y2)
        g2 = g.reshape(M, C_)
        g2 = g2 if g2.is_contiguous() else g2.contiguous()
        masked = d2[0] > 0.0 or dp[0] > 0.0
        dz2 = torch.empty_like(g2) if masked else g2
        dz1 = torch.empty(M, Hd, dtype=y2.dtype, device=y2.device)
        dy = torch.empty(M, C_, dtype=y2.dtype, device=y2.device)
        pk = pack_for(y2.device)
        a = L.Mlp2BwdArgs()
        a.dtype, a.M, a.C, a.Hd = K.dt_code(y2.dtype), M, C_, Hd
        a.g, a.ldg, a.z1 = g2.data_ptr(), C_, z1.data_ptr()
        a.w1_rm, a.w2_rm = pk.get(w1, y2.dtype)[0].data_ptr(), pk.get(w2, y2.dtype)[0].data_ptr()
        a.drop1_p, a.drop1_site, a.drop2_p, a.drop2_site = float(d1[0]), int(d1[1]), float(d2[0]), int(d2[1])
        a.dp_p, a.dp_site, a.dp_rows = float(dp[0]), int(dp[1]), int(dp[2])
        a.rng = rt.rng.data_ptr()
        a.dz2 = dz2.data_ptr() if masked else None
        a.dz1, a.dy, a.lddy = dz1.data_ptr(), dy.data_ptr(), C_
        DeferDW.arm()
        L.check(L.load().qavit_mlp2_bwd(C.byref(a), K.stream()), "mlp2_bwd")
        # dW2 += dz2^T h1, db2 += colsum(dz2);  dW1 += dz1^T y, db1 += colsum(dz1)
        for dz, xin, w, b, n, kd in ((dz2, h1, w2, b2, C_, Hd), (dz1, y2, w1, b1, Hd, C_)):
            wbuf, _ = grad_sink(w)
            bbuf, _ = grad_sink(b)
            if wbuf is None and bbuf is None:
                continue
            if wbuf is None:
                wbuf = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
            K.gemm_tn(dz, xin, wbuf, M, n, kd, n, kd, kd, bbuf)
        return (dy.reshape(yshape) if ctx.needs_input_grad[0] else None), (g if ctx.needs_input_grad[1] else None), None, None, None, None, None


class FanOutFn(Function):
    """k aliases of x; backward sums the k incoming gradients in ONE kernel (autograd's own fan-in is k-1 pairwise adds).
    Use where a tensor feeds several consumers: ``a, b, c = FanOutFn.apply(x, 3)``."""

    @staticmethod
    def forward(ctx, x, k):
        ctx.k = k
        ctx.set_materialize_grads(False)                   # an alias nobody differentiated through arrives as None, not as zeros
        return tuple(x.view_as(x) for _ in range(k))

    @staticmethod
    def backward(ctx, *grads):
        gs = [g for g in grads if g is not None]
        if not gs:
            return None, None
        if len(gs) == 1:
            return gs[0], None
        gs = [g if g.is_contiguous() else g.contiguous() for g in gs]
        vec = 8 if gs[0].dtype == torch.bfloat16 else 4
        ok = gs[0].is_cuda and gs[0].dtype in (torch.bfloat16, torch.float32) and gs[0].numel() % vec == 0 and len(gs) <= 8 \
            and all(g.dtype == gs[0].dtype and g.shape == gs[0].shape and g.data_ptr() % 16 == 0 for g in gs)
        if not ok:
            out = gs[0]
            for g in gs[1:]:
                out = out + g
            return out, None
        out = torch.empty_like(gs[0])
        K.sum_k(gs, out)
        return out, None


class GateMixFn(Function):
    """y = t + sigmoid(g) * r -- SplitFusion's gated add (csrc/runtime.hip: gate_mix), one kernel each way."""

    @staticmethod
    def forward(ctx, t, r, g):
        K._require_cuda(t, r)
        t, r, g = t.contiguous(), r.contiguous(), g.contiguous()
        y = torch.empty_like(t)
        K.gate_mix_fwd(t, r, g, y)
        ctx.save_for_backward(r, g)
        return y

    @staticmethod
    def backward(ctx, dy):
        r, g = ctx.saved_tensors
        dy = dy.contiguous()
        dr, dg = torch.empty_like(r), torch.empty_like(g)
        K.gate_mix_bwd(dy, r, g, dr, dg)
        return dy, dr, dg


class Mix2Fn(Function):
    """y = s0*a + s1*b, s = softmax(fw) -- SplitFusion's learnable blend (csrc/runtime.hip: mix2)."""

    @staticmethod
    def forward(ctx, a, b, fw):
        K._require_cuda(a, fw)
        a, b = a.contiguous(), b.contiguous()
        y = torch.empty_like(a)
        K.mix2_fwd(a, b, fw.detach(), y)
        ctx.save_for_backward(a, b, fw)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, b, fw = ctx.saved_tensors
        dy = dy.contiguous()
        da, db = torch.empty_like(a), torch.empty_like(b)
        fbuf, fret = grad_sink(fw)
        K.mix2_bwd(dy, a, b, fw.detach(), da, db, fbuf)
        return da, db, fret


class Mix3Fn(Function):
    """y = s0*a + s1*(t + dropout(h)), s = softmax(fw) -- SplitFusion's blend with its second operand built in the kernel
    (csrc/runtime.hip: mix3): no dropout / add launches, backward writes (da, dt, dh) in one pass."""

    @staticmethod
    def forward(ctx, a, t, h, fw, drop):
        K._require_cuda(a, fw)
        rt = _rt(a)
        a, t, h = a.contiguous(), t.contiguous(), h.contiguous()
        y = torch.empty_like(a)
        K.mix3_fwd(a, t, h, fw.detach(), y, drop, rt.rng)
        ctx.save_for_backward(a, t, h, fw)
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        a, t, h, fw = ctx.saved_tensors
        rt = _rt(a)
        dy = dy.contiguous()
        da, dt, dh = torch.empty_like(a), torch.empty_like(t), torch.empty_like(h)
        fbuf, fret = grad_sink(fw)
        K.mix3_bwd(dy, a, t, h, fw.detach(), da, dt, dh, fbuf, ctx.drop, rt.rng)
        return da, dt, dh, fret, None


class Mix3LayerNormFn(Function):
    """LayerNorm(s0*a + s1*(t + dropout(h))) -- SplitFusion's blend and its final norm (HQAViT_CIFAR100.py:953-965) as one autograd node and
    one launch each way (qavit_mix3_ln_fwd / _bwd): `mixed` is written once and read once (by the backward), its gradient never leaves
    the registers."""

    @staticmethod
    def forward(ctx, a, t, h, fw, drop, g, b, eps):
        K._require_cuda(a, fw)
        rt = _rt(a)
        Cc = a.shape[-1]
        rows = a.numel() // Cc
        mixed = torch.empty_like(a)
        y = torch.empty_like(a)
        mean = torch.empty(rows, dtype=torch.float32, device=a.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=a.device)
        K.mix3_ln_fwd(a, t, h, fw.detach(), drop, rt.rng, mixed, g.detach(), b.detach(), eps, y, mean, rstd, rows, Cc)
        ctx.save_for_backward(a, t, h, fw, mixed, g, b, mean, rstd)
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        a, t, h, fw, mixed, g, b, mean, rstd = ctx.saved_tensors
        rt = _rt(a)
        Cc = a.shape[-1]
        rows = a.numel() // Cc
        dy = dy.contiguous()
        da, dt, dh = torch.empty_like(a), torch.empty_like(t), torch.empty_like(h)
        fbuf, fret = grad_sink(fw)
        gbuf, _ = grad_sink(g)
        bbuf, _ = grad_sink(b)
        DeferDW.arm()
        K.mix3_ln_bwd(dy, a, t, h, fw.detach(), ctx.drop, rt.rng, mixed, g.detach(), mean, rstd, da, dt, dh, fbuf, gbuf, bbuf, rows, Cc)
        return da, dt, dh, fret, None, None, None, None


class GateMix3LayerNormFn(Function):
    """LayerNorm(s0*(t + sigmoid(gl)*r) + s1*(t + dropout(h))) -- SplitFusion's gate, blend and final norm (HQAViT_CIFAR100.py:945-965) as
    one autograd node and one launch each way (qavit_gate_mix3_ln_fwd / _bwd).  The gated sum is never written; the backward returns ONE
    gradient for t where the separate nodes returned two (s0*dm through the gate, s1*dm from the blend) for the fan-in sum to add."""

    @staticmethod
    def forward(ctx, t, r, gl, h, fw, drop, g, b, eps):
        K._require_cuda(t, fw)
        rt = _rt(t)
        Cc = t.shape[-1]
        rows = t.numel() // Cc
        mixed = torch.empty_like(t)
        y = torch.empty_like(t)
        mean = torch.empty(rows, dtype=torch.float32, device=t.device)
        rstd = torch.empty(rows, dtype=torch.float32, device=t.device)
        K.gate_mix3_ln_fwd(t, r, gl, h, fw.detach(), drop, rt.rng, mixed, g.detach(), b.detach(), eps, y, mean, rstd, rows, Cc)
        ctx.save_for_backward(t, r, gl, h, fw, mixed, g, b, mean, rstd)
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        t, r, gl, h, fw, mixed, g, b, mean, rstd = ctx.saved_tensors
        rt = _rt(t)
        Cc = t.shape[-1]
        rows = t.numel() // Cc
        dy = dy.contiguous()
        dt, dr, dg, dh = torch.empty_like(t), torch.empty_like(r), torch.empty_like(gl), torch.empty_like(h)
        fbuf, fret = grad_sink(fw)
        gbuf, _ = grad_sink(g)
        bbuf, _ = grad_sink(b)
        DeferDW.arm()
        K.gate_mix3_ln_bwd(dy, t, r, gl, h, fw.detach(), ctx.drop, rt.rng, mixed, g.detach(), mean, rstd, dt, dr, dg, dh, fbuf, gbuf, bbuf, rows, Cc)
        return dt, dr, dg, dh, fret, None, None, None, None


class ScaleAddFn(Function):
    """y = x + droppath(gamma * u)"""

    @staticmethod
    def forward(ctx, x, u, gamma, dp):
        rt = _rt(x)
        Cc = x.shape[-1]
        x = x.contiguous()
        u = u.contiguous()
        rows = x.numel() // Cc
        y = torch.empty_like(x)
        K.scale_add_fwd(x, u, None if gamma is None else gamma.detach(), y, rows, Cc, dp, rt.rng)
        ctx.save_for_backward(u, gamma)
        ctx.dp = dp
        return y

    @staticmethod
    def backward(ctx, dy):
        u, gamma = ctx.saved_tensors
        rt = _rt(u)
        Cc = u.shape[-1]
        rows = u.numel() // Cc
        dy = dy.contiguous()
        du = torch.empty_like(u)
        gbuf, gret = grad_sink(gamma)
        K.scale_add_bwd(dy, u, None if gamma is None else gamma.detach(), du, gbuf, rows, Cc, ctx.dp, rt.rng)
        return dy, du, gret, None


class ChanScaleAddFn(Function):
    """y = x + droppath(gamma[c] * u): ConvNeXt layer scale (HQAViTv2_CIFAR100.py:744-748)."""

    @staticmethod
    def forward(ctx, x, u, gamma, dp):
        rt = _rt(x)
        Cc = x.shape[-1]
        x = x.contiguous()
        u = u.contiguous()
        rows = x.numel() // Cc
        y = torch.empty_like(x)
        K.chan_scale_add_fwd(x, u, gamma.detach(), y, rows, Cc, dp, rt.rng)
        ctx.save_for_backward(u, gamma)
        ctx.dp = dp
        return y

    @staticmethod
    def backward(ctx, dy):
        u, gamma = ctx.saved_tensors
        rt = _rt(u)
        Cc = u.shape[-1]
        rows = u.numel() // Cc
        dy = dy.contiguous()
        du = torch.empty_like(u)
        gbuf, gret = grad_sink(gamma)
        if gbuf is None:
            gbuf = torch.zeros_like(gamma, dtype=torch.float32)
        K.chan_scale_add_bwd(dy, u, gamma.detach(), du, gbuf, rows, Cc, ctx.dp, rt.rng)
        return dy, du, _ret(gret, gamma), None


class DropoutFn(Function):
    @staticmethod
    def forward(ctx, x, p, site):
        rt = _rt(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        K.dropout(x, y, p, site, rt.rng)
        ctx.ps = (p, site)
        return y

    @staticmethod
    def backward(ctx, dy):
        rt = _rt(dy)
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        K.dropout(dy, dx, ctx.ps[0], ctx.ps[1], rt.rng)
        return dx, None, None


class BatchNormFn(Function):
    """nn.BatchNorm2d (+ exact GELU when ``act``) on channel-last rows [M, C] -- csrc/bnorm.hip.  Running statistics
    are updated in place by the forward kernel in training mode (the caller bumps ``num_batches_tracked``).

    ``BatchNormFn.sync`` (set by parallel.DataParallel(bn_sync="exact")): callable(stats fp32 [2C]) -> world size that
    SUM-all-reduces the column statistics in place.  Training-mode forward and backward then run as statistics pass ->
    all-reduce -> apply pass over the GLOBAL batch (SyncBN; SURVEY.md 8e exception 2), which makes an N-rank step equal
    the single-process step on the concatenated batch."""
    sync = None

    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, momentum, eps, act, training):
        K._require_cuda(x, weight)
        x = x.contiguous()
        M, Cc = x.shape
        y = torch.empty_like(x)
        dev = x.device
        save_mean = torch.empty(Cc, dtype=torch.float32, device=dev) if training else None
        save_rstd = torch.empty(Cc, dtype=torch.float32, device=dev) if training else None
        ws = torch.empty(3 * Cc, dtype=torch.float32, device=dev) if training else None
        sync = BatchNormFn.sync if training else None
        m_total = 0
        if sync is not None:
            K.bn_fwd(x, y, M, Cc, weight, bias, running_mean, running_var, momentum, eps, act, save_mean, save_rstd, ws, training, phase=1)
            m_total = M * int(sync(ws[:2 * Cc]))             # the pivot (ws[2C:]) is the replicated running mean: identical on all ranks
            K.bn_fwd(x, y, M, Cc, weight, bias, running_mean, running_var, momentum, eps, act, save_mean, save_rstd, ws, training, phase=2, m_total=m_total)
        else:
            K.bn_fwd(x, y, M, Cc, weight, bias, running_mean, running_var, momentum, eps, act, save_mean, save_rstd, ws, training)
        if not training:                                   # eval backward: the running statistics are constants
            save_mean = running_mean.detach().clone()
            save_rstd = torch.rsqrt(running_var.detach() + eps)
        ctx.meta = (M, Cc, bool(act), bool(training), m_total)
        ctx.save_for_backward(x, weight, bias, save_mean, save_rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, weight, bias, save_mean, save_rstd = ctx.saved_tensors
        M, Cc, act, training, m_total = ctx.meta
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        gbuf, gret = grad_sink(weight)
        bbuf, bret = grad_sink(bias)
        ws = torch.empty(2 * Cc, dtype=torch.float32, device=x.device)
        sync = BatchNormFn.sync if (training and m_total > 0) else None
        if sync is not None:
            K.bn_bwd(dy, x, M, Cc, weight, bias, save_mean, save_rstd, act, training, dx, gbuf, bbuf, ws, phase=1)
            local = ws.clone()                              # dgamma / dbeta stay this rank's sums; the gradient all-reduce adds the rest
            sync(ws)
            K.bn_bwd(dy, x, M, Cc, weight, bias, save_mean, save_rstd, act, training, dx, gbuf, bbuf, ws, phase=2, m_total=m_total, ws_param=local)
        else:
            K.bn_bwd(dy, x, M, Cc, weight, bias, save_mean, save_rstd, act, training, dx, gbuf, bbuf, ws)
        return dx, _ret(gret, weight), _ret(bret, bias), None, None, None, None, None, None


class CrossEntropyFn(Function):
    """nn.CrossEntropyLoss(label_smoothing) -- optionally the MixUp / CutMix pair loss lam * CE(y_a) + (1 - lam) * CE(y_b) with a
    device-resident lam -- as one kernel that also leaves the logit gradient (csrc/loss.hip); backward scales it by the incoming
    gradient."""

    @staticmethod
    def forward(ctx, logits, y_a, y_b, lam, ls):
        K._require_cuda(logits)
        lg = logits.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=lg.device)
        need = logits.requires_grad
        d = torch.empty_like(lg) if need else None
        K.ce_label_smooth(lg, y_a, y_b, lam, ls, loss, d)
        ctx.save_for_backward(d)
        return loss

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        return (d * g.to(d.dtype)) if d is not None else None, None, None, None, None


def cross_entropy(logits, y, label_smoothing=0.0, y_b=None, lam=None):
    return CrossEntropyFn.apply(logits, y, y_b, lam, float(label_smoothing))


def dropout(x, p, site, training):
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, p, site)


@torch.no_grad()
def patchify(img, patch, dtype):
    """[B,C,H,W] fp32 image -> [B*(H/p)*(W/p), C*p*p] rows in the compute dtype (no gradient to pixels)."""
    B, Cin, H, W = img.shape
    img = img.contiguous().float()
    cols = torch.empty(B * (H // patch) * (W // patch), Cin * patch * patch, dtype=dtype, device=img.device)
    K.patchify(img, cols, B, Cin, H, W, patch)
    return cols


class _Snapshot2Fn(Function):
    """(k, v) -> (k.clone(), v.clone()) as one kernel; gradients pass straight through (what ``clone`` does)."""

    @staticmethod
    def forward(ctx, k, v):
        return K.copy2(k, v)

    @staticmethod
    def backward(ctx, dk, dv):
        return dk, dv


class BankProj2Fn(Function):
    """(sh_k, sh_v) = (Linear_k(bank_k), Linear_v(bank_v)): the batch-invariant K / V projections of the bank in the cross-attention
    and channel-group branches (HQAViT_CIFAR100.py:576-577, :613-616; the reference applies the Linear to the EXPANDED bank).  One
    grouped launch forward, one backward (two M = 16 GEMMs each were a launch of their own), the weight gradients see the
    forward-time bank (``snap``: the copy the last bank write left, else taken here) and the bank rows' own gradient is accumulated
    into the parameters' ``.grad`` by the input-gradient GEMM's residual epilogue -- no snapshot node, no AccumulateGrad adds."""

    @staticmethod
    def forward(ctx, gk, gv, snap_k, snap_v, wk, bk, wv, bv):
        K._require_cuda(gk, wk)
        S, Cc = gk.shape[-2], gk.shape[-1]
        need = any(ctx.needs_input_grad)
        if need and snap_k is None:
            snap_k, snap_v = K.copy2(gk, gv)
        xk, xv = gk.detach().reshape(S, Cc), gv.detach().reshape(S, Cc)
        pk = pack_for(gk.device)
        Wk, _ = pk.get(wk, gk.dtype)
        Wv, _ = pk.get(wv, gk.dtype)
        nk, nv = wk.shape[0], wv.shape[0]
        yk = torch.empty(S, nk, dtype=gk.dtype, device=gk.device)
        yv = torch.empty(S, nv, dtype=gk.dtype, device=gk.device)
        K.gemm_nt_grouped([K.gemm_nt(xk, Wk, yk, S, nk, Cc, Cc, Cc, nk, None if bk is None else bk.detach(), build_only=True),
                           K.gemm_nt(xv, Wv, yv, S, nv, Cc, Cc, Cc, nv, None if bv is None else bv.detach(), build_only=True)])
        if need:
            ctx.save_for_backward(snap_k, snap_v, wk, bk, wv, bv)
            ctx.bank = (gk, gv)                             # the parameters themselves (their values are mutated in place later: not "saved")
            ctx.set_materialize_grads(False)                # a branch that defers this backward (DeferredBank) sends no gradient here
            ctx.rec = (gk, gv, snap_k, snap_v, wk, bk, wv, bv)
        return yk, yv

    @staticmethod
    def backward(ctx, dk, dv):
        if dk is None and dv is None:
            return (None,) * 8
        snap_k, snap_v, wk, bk, wv, bv = ctx.saved_tensors
        gk, gv = ctx.bank
        S = gk.shape[-2]
        nk, nv = wk.shape[0], wv.shape[0]
        dk2 = torch.zeros(S, nk, dtype=gk.dtype, device=gk.device) if dk is None else dk.reshape(S, nk).contiguous()
        dv2 = torch.zeros(S, nv, dtype=gk.dtype, device=gk.device) if dv is None else dv.reshape(S, nv).contiguous()
        DeferDW.arm()
        outs = _bank_proj2_backward([((gk, gv, snap_k, snap_v, wk, bk, wv, bv), dk2, dv2)])
        return outs[0], outs[1], None, None, None, None, None, None


def _bank_proj2_backward(items):
    """Backward of BankProj2Fn for a list of (record, d sh_k [S, nk], d sh_v [S, nv]): ONE grouped launch for all the bank-row
    gradients (residual epilogue: they accumulate into the parameters' ``.grad``), the weight gradients join the deferred queue.
    -> the autograd returns for (bank_k, bank_v) of the LAST item (None where the gradient went into ``.grad``)."""
    probs = []
    outs = [None, None]
    tn = []
    for (gk, gv, snap_k, snap_v, wk, bk, wv, bv), dk2, dv2 in items:
        S, Cc = gk.shape[-2], gk.shape[-1]
        nk, nv = wk.shape[0], wv.shape[0]
        pk = pack_for(gk.device)
        outs = [None, None]
        for i, (g_, d2, w, n) in enumerate(((gk, dk2, wk, nk), (gv, dv2, wv, nv))):
            if not g_.requires_grad:
                continue
            _, Wt = pk.get(w, gk.dtype)
            sink, ret = grad_sink(g_)
            dst = sink.reshape(S, Cc)
            # dst = d2 @ W + dst: the residual epilogue reads and writes the same element in one thread
            probs.append(K.gemm_nt(d2, Wt, dst, S, Cc, n, n, Wt.shape[1], Cc, None, R=dst, ldr=Cc, build_only=True))
            outs[i] = ret
        for (x_s, d2, w, b, n) in ((snap_k, dk2, wk, bk, nk), (snap_v, dv2, wv, bv, nv)):
            wbuf, _ = grad_sink(w)
            bbuf, _ = grad_sink(b)
            if wbuf is None and bbuf is None:
                continue
            if wbuf is None:
                wbuf = torch.zeros(w.shape, dtype=torch.float32, device=w.device)
            tn.append((d2, x_s.reshape(S, Cc), wbuf, S, n, Cc, n, Cc, Cc, bbuf))
    if probs:
        _grouped_by_destination(probs)                       # several projections accumulate into the same bank parameter
    for a in tn:
        K.gemm_tn(*a)
    return outs


def _grouped_by_destination(probs):
    """gemm_nt_grouped over problems whose residual epilogue accumulates into a shared destination: problems with the same output
    pointer must not run in the same launch (read-modify-write without atomics), so they are dealt into successive launches."""
    rounds = []
    for pr in probs:
        key = int(pr.C)
        for r in rounds:
            if key not in r[0]:
                r[0].add(key); r[1].append(pr)
                break
        else:
            rounds.append(({key}, [pr]))
    for _, plist in rounds:
        K.gemm_nt_grouped(plist)


_BANK_DEFER = os.environ.get("QAVIT_BANK_DEFER", "1") != "0"


class DeferredBank:
    """The backward of every BankProj2Fn of a pass, run once when the pass ends (DeferDW._launch).  The fused branch kernels leave
    the projected rows' gradient as partial rows; nothing but the bank parameters' and the projection weights' ``.grad`` depends on
    them, so instead of (zero fill, reduce, grouped dX launch) on the critical path of each of the 16 cross-attention and
    channel-group branches, the partial rows are folded by the pass's single reduce launch into slots of one zeroed pool and the 32
    bank-row GEMMs run as a handful of grouped launches (problems that accumulate into the same parameter in successive ones)."""
    queue = []          # (record, both[2, S, n] fp32)
    _pools = {}         # shape -> [pool tensor, used]

    @staticmethod
    def record_of(sh_k, sh_v):
        if not (_BANK_DEFER and K.DeferredLN.enabled and K.DeferredLN.ON):
            return None
        rec = getattr(sh_k, "_qavit_bank_rec", None)
        if rec is None or rec is not getattr(sh_v, "_qavit_bank_rec", None):
            return None
        if not (sh_k.requires_grad and sh_v.requires_grad and sh_k.shape == sh_v.shape and rec[0].is_leaf and rec[1].is_leaf):
            return None
        return rec

    @classmethod
    def slot(cls, shape, device):
        key = (tuple(shape), str(device))
        ent = cls._pools.get(key)
        if ent is None or ent[1] >= ent[0].shape[0]:
            ent = [torch.zeros((32, 2) + tuple(shape), dtype=torch.float32, device=device), 0]
            cls._pools[key] = ent
        ent[1] += 1
        return ent[0][ent[1] - 1]

    @classmethod
    def run(cls):
        q, cls.queue = cls.queue, []
        cls._pools = {}
        _bank_proj2_backward([(rec, both[0], both[1]) for rec, both in q])


def bank_proj2(bank, snap, lin_k, lin_v):
    """-> (Linear_k(bank.global_k), Linear_v(bank.global_v)) as [S, n] matrices; see BankProj2Fn."""
    sk, sv = snap if snap is not None else (None, None)
    yk, yv = BankProj2Fn.apply(bank.global_k, bank.global_v, sk, sv, lin_k.weight, lin_k.bias, lin_v.weight, lin_v.bias)
    fn = yk.grad_fn
    rec = getattr(fn, "rec", None) if fn is not None else None
    if rec is not None:                                     # lets a fused branch backward defer this node's backward (DeferredBank)
        yk._qavit_bank_rec = rec
        yv._qavit_bank_rec = rec
    return yk, yv


class _SnapshotGivenFn(Function):
    """_Snapshot2Fn whose copies already exist (written by the bank write that produced the current rows)."""

    @staticmethod
    def forward(ctx, k, v, k_copy, v_copy):
        return k_copy.view_as(k), v_copy.view_as(v)

    @staticmethod
    def backward(ctx, dk, dv):
        return dk, dv, None, None


def bank_snapshot(bank, snap=None):
    """The bank rows as the reference's ``Linear`` on the EXPANDED bank sees them: a copy taken now, so the weight gradient
    is computed against the forward-time bank although ``GlobalTokenBank.write`` mutates the parameter in place later in
    the same forward (HQAViT_CIFAR100.py:576-577).  ``snap``: that copy, if the last bank write already made it."""
    if snap is not None and torch.is_grad_enabled():
        return _SnapshotGivenFn.apply(bank.global_k, bank.global_v, snap[0], snap[1])
    return _Snapshot2Fn.apply(bank.global_k, bank.global_v)


@torch.no_grad()
def bank_write(tokens, norm_g, norm_b, bank, mode, sync=None, want_snap=False):
    """GlobalTokenBank.write(norm(tokens)) -- see csrc/bank.hip.  ``bank`` is the GlobalTokenBank module.
    ``sync(acc)`` (optional) all-reduces the [S,C] batch sum across data-parallel ranks and returns the
    global batch size divisor."""
    B, N, Cc = tokens.shape
    rt = _rt(tokens)
    S = bank.bank_size
    tokens = tokens.contiguous()
    acc = rt.workspace("bank_acc", S * Cc + 1, zero=True)     # zero on entry; bank_apply leaves it zero again (+1: its ticket word)
    n_ws = K.bank_ws_floats(B, N, Cc, S)                      # = workgroups of the statistics kernel * S * C
    ws = rt.workspace("bank_ws", n_ws)
    fold_in_apply = sync is None and Cc % 4 == 0          # single GPU: apply folds the partials itself (no reduce launch)
    fix = None
    if rt.pending_fix is not None:                          # the branch that produced `tokens` left its NaN rule to this launch
        fix, fout, _keep = rt.pending_fix
        rt.pending_fix = None
        if fout.data_ptr() != tokens.data_ptr():
            raise RuntimeError("the deferred NaN rule belongs to another tensor than the one being written to the bank")
    K.bank_stats(tokens, norm_g, norm_b, bank.write_norm.weight, bank.write_norm.bias, bank.write_gate.weight, bank.write_gate.bias,
                 None if fold_in_apply else acc, ws, B, N, Cc, S, 1e-5, fix=fix)
    total = B
    if sync is not None:
        total = sync(acc[: S * Cc], B)
    snap = (torch.empty_like(bank.global_k.data), torch.empty_like(bank.global_v.data)) if want_snap else None
    K.bank_apply(acc, bank.write_compression.weight, bank.write_compression.bias, bank.global_k.data, bank.global_v.data,
                 getattr(bank, "update_count", None), S, Cc, 1.0 / float(total), mode,
                 ws if fold_in_apply else None, n_ws // (S * Cc) if fold_in_apply else 0, snap=snap)
    return snap
