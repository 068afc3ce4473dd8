"""Raw (non-autograd) Python wrappers over the C-ABI: tensors in, kernels enqueued on torch's current
stream.  PyTorch is plumbing here -- device memory, streams -- the arithmetic is in libqavit_hip.so."""
import os
import ctypes as C
from typing import Optional

import torch

from . import lib as L


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def dt_code(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return L.F32
    if dtype == torch.bfloat16:
        return L.BF16
    raise TypeError(f"qavit kernels support float32 and bfloat16 activations, got {dtype}")


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("qavit HIP kernels need tensors on the GPU (there is no CPU fallback)")


class Stamps:
    """Diagnostic (QAVIT_STAMPS=1; tools/chain_stamps.py): named wall-clock readings taken by one-lane kernels on whatever stream is
    current, so a captured step records where its chains are without a profiler attached.  Off: ``mark`` is a no-op."""
    enabled = os.environ.get("QAVIT_STAMPS", "0") != "0"
    names: list = []
    buf = None

    @classmethod
    def mark(cls, name: str):
        if not cls.enabled:
            return
        if cls.buf is None:
            cls.buf = torch.zeros(128, dtype=torch.int64, device="cuda")
        if name not in cls.names:
            cls.names.append(name)
        L.check(L.load().qavit_stamp(cls.buf.data_ptr() + 8 * cls.names.index(name), stream()), "stamp")

    @classmethod
    def read(cls):
        """{name: microseconds since the earliest stamp} of the last pass over each mark."""
        v = cls.buf[:len(cls.names)].cpu().tolist()
        t0 = min(v)
        return {n: (t - t0) / 100.0 for n, t in zip(cls.names, v)}


class Runtime:
    """Per-device state: RNG words, NaN flags, dropout-site ids, scratch workspaces."""
    _inst = {}

    def __init__(self, device):
        self.device = device
        self.rng = torch.tensor([0x5EED, 0], dtype=torch.int64, device=device)
        self.nan_flag = torch.zeros(2, dtype=torch.int32, device=device)       # [flag, guard arrival ticket]
        self.pending_fix = None        # (lib.NanFix, out tensor, keepalive): a fused branch's NaN rule waiting for the bank write that follows it
        self._ws = {}
        self._tbl = {}
        self.nan_guard = True

    @classmethod
    def get(cls, device) -> "Runtime":
        key = torch.device(device).index or 0
        if key not in cls._inst:
            cls._inst[key] = Runtime(torch.device("cuda", key))
        return cls._inst[key]

    @classmethod
    def drop_pending_fix(cls):
        """A call failed between a fused branch deferring its NaN rule and the launch that would have consumed it (an OOM, a refused
        operand, an interrupt): forget the deferral on every device, or every later forward raises 'never consumed'."""
        for r in cls._inst.values():
            r.pending_fix = None

    def seed(self, s: int):
        self.rng[0] = int(s)
        self.rng[1] = 0

    def advance(self):
        L.check(L.load().qavit_rng_advance(self.rng.data_ptr(), stream()), "rng_advance")

    def workspace(self, name: str, n_floats: int, zero: bool = False) -> torch.Tensor:
        """Named scratch buffer; ``zero`` = zero-filled at creation (for buffers whose users leave them zero again)."""
        w = self._ws.get(name)
        if w is None or w.numel() < n_floats:
            alloc = torch.zeros if zero else torch.empty
            w = alloc(max(int(n_floats), 1), dtype=torch.float32, device=self.device)
            self._ws[name] = w
        return w

    def table(self, key, builder) -> torch.Tensor:
        t = self._tbl.get(key)
        if t is None:
            t = torch.tensor(builder(), dtype=torch.int32, device=self.device)
            self._tbl[key] = t
        return t


_site_counter = [0]


def new_site() -> int:
    """Unique id of a dropout / drop-path site (mixed into the RNG key)."""
    _site_counter[0] += 1
    return _site_counter[0]


# ---------------------------------------------------------------------------------------------------
# GEMMs
# ---------------------------------------------------------------------------------------------------
def gemm_nt(A, B, Cout, M, N, K, lda, ldb, ldc, bias=None, *, a_mode=0, ln=None, ln_stats=None,
            bwd=None, Z=None, act=0, drop=(0.0, 0), scale=1.0, dp=(0.0, 0, 1), R=None, ldr=0, rng=None,
            A_ptr=None, B_ptr=None, C_ptr=None, build_only=False, A2=None, lda2=0, a2_k0=0, lnbwd=None):
    """C[M,N] = epi(pro(A)[M,K] @ B[N,K]^T + bias).  A/B/Cout are tensors used for dtype/liveness; *_ptr
    override the base address (column-offset views).  ``build_only`` returns the argument struct for gemm_nt_grouped.
    ``lnbwd`` = dict(x, mean, rstd, gamma, dgamma, dbeta[, adds]): the LayerNorm-backward epilogue (qavit_gemm_args.e_x; check
    gemm_nt_lnbwd_ok first) -- the product is the gradient of LayerNorm(x)'s output, Cout receives the gradient of x (+ R); the
    LayerNorm parameter gradients leave as partial rows when a backward pass has armed DeferredLN, else as float atomics."""
    a = L.GemmArgs()
    a.dtype = dt_code(A.dtype)
    a.M, a.N, a.K = M, N, K
    a.A, a.lda = (A_ptr or A.data_ptr()), lda
    a.B, a.ldb = (B_ptr or B.data_ptr()), ldb
    a.C, a.ldc = (C_ptr or Cout.data_ptr()), ldc
    a.bias = _p(bias)
    a.a_mode = a_mode
    a.a_scale = 1.0
    if a_mode in (1, 3):                      # 1: ln_stats are inputs (row_stats ran); 3: the call computes and writes them
        g, b_, eps = ln
        a.ln_gamma, a.ln_beta, a.ln_eps = g.data_ptr(), b_.data_ptr(), eps
        if ln_stats is not None:
            a.ln_mean, a.ln_rstd = ln_stats[0].data_ptr(), ln_stats[1].data_ptr()
    if a_mode == 2:
        a.a_Z = _p(bwd.get("Z"))
        a.a_ldz = bwd.get("ldz", 0)
        a.a_act = bwd.get("act", 0)
        a.a_drop_p, a.a_drop_site = bwd.get("drop", (0.0, 0))
        a.a_dp_p, a.a_dp_site, a.a_dp_rows = bwd.get("dp", (0.0, 0, 1))
        a.a_scale = bwd.get("scale", 1.0)
        a.a_out = _p(bwd.get("out"))
        a.a_ldo = bwd.get("ldo", 0)
    if Z is not None:
        a.Z, a.ldz = Z.data_ptr(), N
    a.act = act
    a.drop_p, a.drop_site = drop
    a.scale = scale
    a.dp_p, a.dp_site, a.dp_rows = dp
    if R is not None:
        a.R, a.ldr = R.data_ptr(), (ldr or N)
    a.rng = _p(rng)
    if A2 is not None:                        # two-source A: columns a2_k0 .. K of the contraction come from A2 (qavit_gemm_args.A2)
        a.A2, a.lda2, a.a2_k0 = A2.data_ptr(), lda2, a2_k0
    ln_parts = None
    if lnbwd is not None:
        a.e_x, a.e_mean, a.e_rstd, a.e_gamma = lnbwd["x"].data_ptr(), lnbwd["mean"].data_ptr(), lnbwd["rstd"].data_ptr(), lnbwd["gamma"].data_ptr()
        adds = [t for t in (lnbwd.get("adds") or ()) if t is not None]
        if len(adds) > 2:
            raise ValueError("gemm_nt: the LayerNorm-backward epilogue takes at most two addends")
        if adds:
            a.e_add0 = adds[0].data_ptr()
        if len(adds) > 1:
            a.e_add1 = adds[1].data_ptr()
        dg, db = lnbwd.get("dgamma"), lnbwd.get("dbeta")
        if (dg is not None or db is not None) and DeferredLN.enabled and DeferredLN.ON:
            n_ = int(L.load().qavit_gemm_nt_lnbwd_parts(M, N, K))
            ln_parts = (torch.empty(n_ * 2 * N, dtype=torch.float32, device=A.device), n_)
            a.e_parts = ln_parts[0].data_ptr()
        else:
            a.e_dgamma, a.e_dbeta = _p(dg), _p(db)
    if build_only:
        return a
    L.check(L.load().qavit_gemm_nt(C.byref(a), stream()), "gemm_nt")
    if ln_parts is not None:
        DeferredLN.push(ln_parts[0], ln_parts[1], N, lnbwd.get("dgamma"), lnbwd.get("dbeta"), keep=(lnbwd["x"],))


def gemm_nt_lnbwd_ok(x, M, N, K, a_mode, *others) -> bool:
    """Can qavit_gemm_nt run its LayerNorm-backward epilogue on this input-gradient GEMM?  (x = the LayerNorm's input [M, N]; ``others``:
    the addends / residual, same shape and dtype)"""
    return (x.dtype == torch.bfloat16 and x.is_contiguous() and x.data_ptr() % 16 == 0
            and all(t is None or (t.dtype == x.dtype and t.is_contiguous() and t.numel() == x.numel() and t.data_ptr() % 16 == 0) for t in others)
            and bool(L.load().qavit_gemm_nt_lnbwd_supported(dt_code(x.dtype), M, N, K, a_mode)))


def gemm_nt_a2_ok(x, x2, M, N, K, k0) -> bool:
    return (x.dtype == torch.bfloat16 == x2.dtype and x.data_ptr() % 16 == 0 and x2.data_ptr() % 16 == 0
            and bool(L.load().qavit_gemm_nt_a2_supported(dt_code(x.dtype), M, N, K, k0)))


def gemm_nt_grouped(args):
    """Independent gemm_nt problems (structs from ``gemm_nt(..., build_only=True)``); same-shape ones share a grid."""
    arr = (L.GemmArgs * len(args))(*args)
    L.check(L.load().qavit_gemm_nt_grouped(arr, len(args), stream()), "gemm_nt_grouped")


class DeferredTN:
    """Queue of weight-gradient GEMMs (see qavit_gemm_tn_grouped).  ``enabled`` is switched on by the autograd layer
    for the duration of a backward pass; queued problems keep their operand tensors alive until ``flush``."""
    enabled = False
    home_stream = None  # raw stream handle of the backward pass that armed the queue; problems queued from other streams are adopted at flush
    queue = []          # (GemmTnArgs, keepalive tuple, torch stream if queued from a foreign stream)
    # ASYNC (opt-in, QAVIT_DW_ASYNC=1): the grouped launches go to a stream of their own, forked off the backward stream every MAX
    # problems and joined when backward ends (or at a data-parallel sync point) -- a graph branch under hipGraph capture.  Measured
    # SLOWER on MI355X (B = 1024 step: 16.3 ms grouped at the end of backward, 17.2-17.5 ms with the weight gradients beside the dX
    # chain): these are chip-filling, atomic-bound kernels, and what they take from the latency-bound chain exceeds what they hide.
    ASYNC = os.environ.get("QAVIT_DW_ASYNC", "0") != "0"
    MAX = int(os.environ.get("QAVIT_DW_QUEUE", "24" if ASYNC else "256"))
    SPLIT_FOREIGN = os.environ.get("QAVIT_DW_SPLIT_FOREIGN", "0") != "0"
    _side = {}
    _dirty = set()

    _ws_bytes = None

    @classmethod
    def _grouped(cls, arr, n):
        """qavit_gemm_tn_grouped_ws with a scratch buffer of this call's own (torch's allocator keeps it stream-ordered, also inside a
        capture): all the problems of the call then share ONE launch through a device-side problem table."""
        lib = L.load()
        if n < 2:
            return L.check(lib.qavit_gemm_tn_grouped(arr, n, stream()), "gemm_tn_grouped")
        if cls._ws_bytes is None:
            cls._ws_bytes = int(lib.qavit_gemm_tn_ws_bytes())
        ws = torch.empty(cls._ws_bytes, dtype=torch.uint8, device="cuda")
        L.check(lib.qavit_gemm_tn_grouped_ws(arr, n, ws.data_ptr(), cls._ws_bytes, stream()), "gemm_tn_grouped")

    @classmethod
    def _adopt_foreign(cls, q):
        """Problems queued from another stream (the lateral CNN path's backward): the launching stream waits for that stream's work
        so far, and the caching allocator is told the operands are read here."""
        cur = None
        seen = set()
        for _, keep, st in q:
            if st is None:
                continue
            if cur is None:
                cur = torch.cuda.current_stream()
            if st.cuda_stream == cur.cuda_stream:
                continue
            if st.cuda_stream not in seen:
                seen.add(st.cuda_stream)
                cur.wait_stream(st)
            for t in keep:
                for u in (t if isinstance(t, tuple) else (t,)):
                    if isinstance(u, torch.Tensor):
                        u.record_stream(cur)

    @classmethod
    def flush(cls, home_only=False):
        """``home_only``: launch only what the arming stream queued and keep the other streams' problems for the final flush (an early
        flush must not make this stream wait for a chain that is still running beside it)."""
        if not cls.queue:
            return
        q, cls.queue = cls.queue, []
        if home_only:
            cls.queue = [e for e in q if e[2] is not None]
            q = [e for e in q if e[2] is None]
            if not q:
                return
        if not cls.ASYNC:
            # this stream's problems first: the launch that has to wait for the other stream should not hold them back
            parts = ([e for e in q if e[2] is None], [e for e in q if e[2] is not None]) if cls.SPLIT_FOREIGN else (q,)
            for part in parts:
                if part:
                    arr = (L.GemmTnArgs * len(part))(*[a for a, _, _ in part])
                    cls._adopt_foreign(part)
                    cls._grouped(arr, len(part))
            return
        arr = (L.GemmTnArgs * len(q))(*[a for a, _, _ in q])
        cls._adopt_foreign(q)
        dev = torch.cuda.current_device()
        side = cls._side.get(dev)
        if side is None:
            side = cls._side[dev] = torch.cuda.Stream(device=dev)
        main = torch.cuda.current_stream(dev)
        side.wait_stream(main)                              # every operand queued so far has been produced
        with torch.cuda.stream(side):
            cls._grouped(arr, len(q))
        for _, keep, _ in q:                                # the caching allocator must not hand these blocks out before the side stream is done
            for t in keep:
                if isinstance(t, torch.Tensor):
                    t.record_stream(side)
                elif isinstance(t, tuple):
                    for u in t:
                        if isinstance(u, torch.Tensor):
                            u.record_stream(side)
        cls._dirty.add(dev)

    @classmethod
    def flush_range(cls, lo: int, hi: int):
        """Launch only the queued problems of THIS stream whose destination (weight or bias gradient) lies in the address range
        [lo, hi) -- a data-parallel sync point needs exactly the gradients of the bucket it is about to reduce; everything else (other
        stages' problems, the lateral stream's) stays queued for the one grouped launch at the end of backward, where stream-K has the
        most problems to balance."""
        if not cls.queue:
            return
        now, keep = [], []
        for e in cls.queue:
            a = e[0]
            c, cs = int(a.C or 0), int(a.colsum or 0)
            hit = e[2] is None and ((lo <= c < hi) or (cs and lo <= cs < hi))
            (now if hit else keep).append(e)
        if not now:
            return
        cls.queue = now
        cls.flush()
        cls.queue = keep + cls.queue

    @classmethod
    def join(cls):
        """The current stream waits for the weight-gradient stream (before anything reads .grad)."""
        if not cls._dirty:
            return
        dev = torch.cuda.current_device()
        if dev in cls._dirty:
            torch.cuda.current_stream(dev).wait_stream(cls._side[dev])
            cls._dirty.discard(dev)


def gemm_tn(A, Bm, Cgrad, M, N, K, lda, ldb, ldc, colsum=None, ln=None, A_ptr=None, B_ptr=None, C_ptr=None, colsum_ptr=None):
    """Cgrad[N,K] += A[M,N]^T @ B[M,K] (fp32), colsum[N] += sum_m A."""
    a = L.GemmTnArgs()
    a.dtype = dt_code(A.dtype)
    a.M, a.N, a.K = M, N, K
    a.A, a.lda = (A_ptr or A.data_ptr()), lda
    a.B, a.ldb = (B_ptr or Bm.data_ptr()), ldb
    a.C, a.ldc = (C_ptr or Cgrad.data_ptr()), ldc
    a.colsum = colsum_ptr or _p(colsum)
    if ln is not None:
        g, b_, mean, rstd = ln
        a.ln_gamma, a.ln_beta, a.ln_mean, a.ln_rstd = g.data_ptr(), b_.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    a.splits = 0
    if DeferredTN.enabled:
        foreign = DeferredTN.home_stream is not None and DeferredTN.home_stream != stream()
        DeferredTN.queue.append((a, (A, Bm, Cgrad, colsum, ln), torch.cuda.current_stream() if foreign else None))
        if len(DeferredTN.queue) >= DeferredTN.MAX:
            DeferredTN.flush()
        return
    L.check(L.load().qavit_gemm_tn(C.byref(a), stream()), "gemm_tn")


# ---------------------------------------------------------------------------------------------------
# LayerNorm
# ---------------------------------------------------------------------------------------------------
def layernorm_fwd(x, y, gamma, beta, eps, rows, Cc, mean, rstd, add=None, add_rows=0, act=0):
    L.check(L.load().qavit_layernorm_fwd(dt_code(x.dtype), x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                         eps, rows, Cc, _p(mean), _p(rstd), _p(add), add_rows, int(act), stream()), "layernorm_fwd")


def row_stats(x, eps, rows, Cc, mean, rstd):
    L.check(L.load().qavit_row_stats(dt_code(x.dtype), x.data_ptr(), eps, rows, Cc, mean.data_ptr(), rstd.data_ptr(), stream()), "row_stats")


def _ptr_arr(ts):
    return (C.c_void_p * len(ts))(*[t.data_ptr() for t in ts])


def row_stats_multi(xs, eps, rows, Cc, means, rstds):
    L.check(L.load().qavit_row_stats_multi(dt_code(xs[0].dtype), len(xs), _ptr_arr(xs), eps, rows, Cc, _ptr_arr(means), _ptr_arr(rstds), stream()),
            "row_stats_multi")


class DeferredLN:
    """LayerNorm parameter gradients of a backward pass: the backward kernels leave per-workgroup partial sums in a workspace
    (``part_ws`` of qavit_layernorm_bwd) and ONE qavit_ln_param_reduce launch adds them all when the pass ends (same life cycle
    as DeferredTN: armed by functional.DeferDW, flushed at the end of backward and at data-parallel sync points)."""
    enabled = False
    ON = os.environ.get("QAVIT_DEFER_LN", "1") != "0"
    queue = []          # (LnReduceDesc, keepalive tuple, torch stream if queued from a foreign stream)

    @classmethod
    def parts_for(cls, x, dy, dx, rows, Cc, cap=1024):
        """Workspace for one LayerNorm backward, or None when the partial-sum path does not apply."""
        if not (cls.enabled and cls.ON) or Cc % 4 or Cc > cap:
            return None
        vec = 4 * x.element_size()
        if (x.data_ptr() | dy.data_ptr() | dx.data_ptr()) % vec:
            return None
        n = L.load().qavit_layernorm_bwd_parts(rows, Cc)
        return torch.empty(n * 2 * Cc, dtype=torch.float32, device=x.device), n

    @classmethod
    def push(cls, ws, n, Cc, dgamma, dbeta, keep=()):
        cls.push_raw(ws.data_ptr(), n, Cc, _p(dgamma), _p(dbeta), 0, (ws, dgamma, dbeta) + tuple(keep))

    @staticmethod
    def desc(parts_ptr, n, Cc, dg_ptr, db_ptr, stride=0):
        d = L.LnReduceDesc()
        d.parts, d.nparts, d.C = parts_ptr, n, Cc
        d.dgamma, d.dbeta, d.stride = dg_ptr, db_ptr, stride
        return d

    @classmethod
    def push_raw(cls, parts_ptr, n, Cc, dg_ptr, db_ptr, stride, keep):
        """Queue ``dst halves += column sums of n partial rows`` (qavit_ln_reduce_desc); reduced now when no backward pass has armed the queue."""
        d = cls.desc(parts_ptr, n, Cc, dg_ptr, db_ptr, stride)
        if not (cls.enabled and cls.ON):
            reduce_now([d])
            return
        foreign = DeferredTN.home_stream is not None and DeferredTN.home_stream != stream()
        cls.queue.append((d, tuple(keep), torch.cuda.current_stream() if foreign else None))

    @classmethod
    def flush(cls, home_only=False):
        if not cls.queue:
            return
        q, cls.queue = cls.queue, []
        if home_only:                                       # see DeferredTN.flush
            cls.queue = [e for e in q if e[2] is not None]
            q = [e for e in q if e[2] is None]
            if not q:
                return
        DeferredTN._adopt_foreign(q)
        arr = (L.LnReduceDesc * len(q))(*[d for d, _, _ in q])
        L.check(L.load().qavit_ln_param_reduce(arr, len(q), stream()), "ln_param_reduce")


def reduce_now(descs):
    arr = (L.LnReduceDesc * len(descs))(*descs)
    L.check(L.load().qavit_ln_param_reduce(arr, len(descs), stream()), "ln_param_reduce")


def layernorm_bwd_multi(dys, xs, gammas, means, rstds, dxs, dgammas, dbetas, rows, Cc):
    parts = [DeferredLN.parts_for(xs[i], dys[i], dxs[i], rows, Cc, cap=512) for i in range(len(xs))]
    if any(p is None for p in parts):
        parts = None
    L.check(L.load().qavit_layernorm_bwd_multi(dt_code(xs[0].dtype), len(xs), _ptr_arr(dys), _ptr_arr(xs), _ptr_arr(gammas), _ptr_arr(means),
                                               _ptr_arr(rstds), _ptr_arr(dxs), _ptr_arr(dgammas), _ptr_arr(dbetas), rows, Cc,
                                               _ptr_arr([p[0] for p in parts]) if parts else None, stream()),
            "layernorm_bwd_multi")
    if parts:
        for i, (ws, n) in enumerate(parts):
            DeferredLN.push(ws, n, Cc, dgammas[i], dbetas[i])


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, Cc, dadd=None, add_rows=0, beta=None, act=0, dres=None):
    part = DeferredLN.parts_for(x, dy, dx, rows, Cc) if (dgamma is not None or dbeta is not None) else None
    L.check(L.load().qavit_layernorm_bwd(dt_code(x.dtype), dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                         rstd.data_ptr(), dx.data_ptr(), _p(dgamma), _p(dbeta), rows, Cc, _p(dadd), add_rows,
                                         _p(beta), int(act), _p(dres), part[0].data_ptr() if part else None, stream()), "layernorm_bwd")
    if part:
        DeferredLN.push(part[0], part[1], Cc, dgamma, dbeta)


def layernorm_bwd_sum_ok(x, dys, Cc) -> bool:
    vec = 4 * x.element_size()
    return (1 <= len(dys) <= 5 and Cc % 4 == 0 and Cc <= 256 and x.dtype in (torch.bfloat16, torch.float32) and x.data_ptr() % vec == 0
            and all(d.dtype == x.dtype and d.is_contiguous() and d.numel() == x.numel() and d.data_ptr() % vec == 0 for d in dys))


def layernorm_bwd_sum(dys, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, Cc, dres=None):
    """LayerNorm backward on dy = sum(dys) (qavit_layernorm_bwd_sum): the k-way gradient sum happens on load."""
    part = DeferredLN.parts_for(x, dys[0], dx, rows, Cc) if (dgamma is not None or dbeta is not None) else None
    L.check(L.load().qavit_layernorm_bwd_sum(dt_code(x.dtype), len(dys), _ptr_arr(dys), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                             dx.data_ptr(), _p(dgamma), _p(dbeta), rows, Cc, _p(dres), part[0].data_ptr() if part else None, stream()),
            "layernorm_bwd_sum")
    if part:
        DeferredLN.push(part[0], part[1], Cc, dgamma, dbeta)


def layernorm_bwd_lin_ok(x, dz, n, Cc) -> bool:
    return (x.dtype == torch.bfloat16 and dz.dtype == torch.bfloat16 and dz.is_contiguous() and dz.data_ptr() % 16 == 0 and x.data_ptr() % 8 == 0
            and bool(L.load().qavit_layernorm_bwd_lin_supported(dt_code(x.dtype), n, Cc)))


def layernorm_bwd_lin(dz, W_ptr, ldw, n, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, Cc, dres=None):
    """dx = LayerNorm_backward(dz @ W) + dres in one launch (qavit_layernorm_bwd_lin); W_ptr = the [n, C] compute-dtype weight rows."""
    part = DeferredLN.parts_for(x, x, dx, rows, Cc) if (dgamma is not None or dbeta is not None) else None
    L.check(L.load().qavit_layernorm_bwd_lin(dt_code(x.dtype), dz.data_ptr(), n, W_ptr, ldw, n, x.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                             rstd.data_ptr(), dx.data_ptr(), _p(dgamma), _p(dbeta), rows, Cc, _p(dres),
                                             part[0].data_ptr() if part else None, stream()), "layernorm_bwd_lin")
    if part:
        DeferredLN.push(part[0], part[1], Cc, dgamma, dbeta)


def ln_dres_ok(x, dres, Cc, dadd=None) -> bool:
    """Can qavit_layernorm_bwd add ``dres`` in its own pass?  (vector path: C % 4 == 0, aligned, same dtype / shape, no dadd)"""
    vec = 4 * x.element_size()
    return (dres is not None and dadd is None and Cc % 4 == 0 and Cc <= 1024 and dres.dtype == x.dtype and dres.numel() == x.numel()
            and dres.is_contiguous() and dres.data_ptr() % vec == 0 and x.data_ptr() % vec == 0)


# ---------------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------------
def attn_args(dtype, mode, G, Nq, Lk, H, D, KC, S, groups_per_b=0, q_rows_per_b=0, k_rows_per_b=0, q_tbl=None, k_tbl=None):
    a = L.AttnArgs()
    a.dtype, a.mode = dt_code(dtype), mode
    a.G, a.Nq, a.L, a.H, a.D, a.KC, a.S = G, Nq, Lk, H, D, KC, S
    a.groups_per_b, a.q_rows_per_b, a.k_rows_per_b = groups_per_b, q_rows_per_b, k_rows_per_b
    a.q_tbl, a.k_tbl = _p(q_tbl), _p(k_tbl)
    return a


def attn_fwd(a):
    L.check(L.load().qavit_attn_fwd(C.byref(a), stream()), "attn_fwd")


def attn_bwd(a):
    L.check(L.load().qavit_attn_bwd(C.byref(a), stream()), "attn_bwd")


def attn_ws_floats(a) -> int:
    return int(L.load().qavit_attn_ws_floats(C.byref(a)))


def branch_supported(kind, T, Cc, H, D, KC, S, Lk) -> bool:
    return bool(L.load().qavit_branch_supported(kind, T, Cc, H, D, KC, S, Lk))


def branch_fwd(a):
    L.check(L.load().qavit_branch_fwd(C.byref(a), stream()), "branch_fwd")


def branch_bwd(a):
    L.check(L.load().qavit_branch_bwd(C.byref(a), stream()), "branch_bwd")


def branch_bwd_parts(B, T=16) -> int:
    return int(L.load().qavit_branch_bwd_parts(int(B), int(T)))


BRANCH_PARTS_FLOATS = 7168            # include/qavit.h QAVIT_BRANCH_PARTS_FLOATS: [dE_k 16x32 | dE_v 16x32 | d sh_k 16x192 | d sh_v 16x192]
BRANCH_PARTS_FLOATS_64 = 9216         # ... _64 (T = 64): the dE slots hold 48 rows


def nan_guard(x, flag):
    L.check(L.load().qavit_nan_guard(dt_code(x.dtype), x.data_ptr(), x.numel(), flag.data_ptr(), stream()), "nan_guard")


# ---------------------------------------------------------------------------------------------------
# token kernels
# ---------------------------------------------------------------------------------------------------
def tl_ok(x, N, M, Cc) -> bool:
    """Can the fused TokenLearner kernels (qavit_tl_fwd / qavit_tl_bwd) take this problem?"""
    return (x.dtype == torch.bfloat16 and x.is_contiguous() and x.data_ptr() % 16 == 0
            and bool(L.load().qavit_tl_supported(dt_code(x.dtype), N, M, Cc)))


def tl_fwd(x, ln_g, ln_b, eps, Wc, bias, p, xc, mean, rstd, B, N, M, Cc):
    """TokenLearner forward in one launch (qavit_tl_fwd): p = softmax_N(Linear(LN(x))), xc = p^T x; Wc = the [M, C] compute-dtype weight."""
    a = L.TlArgs()
    a.x, a.ln_g, a.ln_b, a.eps = x.data_ptr(), ln_g.data_ptr(), ln_b.data_ptr(), float(eps)
    a.W, a.bias = Wc.data_ptr(), _p(bias)
    a.p, a.xc, a.mean, a.rstd = p.data_ptr(), xc.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    a.B, a.N, a.M, a.C = B, N, M, Cc
    L.check(L.load().qavit_tl_fwd(C.byref(a), stream()), "tl_fwd")


def tl_bwd(dxc, x, p, mean, rstd, ln_g, ln_b, Wc, dx, dW, dbias, dgamma, dbeta, B, N, M, Cc):
    """TokenLearner backward in one launch (qavit_tl_bwd): dx, and the score Linear's / LayerNorm's parameter gradients as one partial
    row per workgroup, folded by the pass's reduce launch (DeferredLN) or at once when no backward pass has armed the queue."""
    lib = L.load()
    n = int(lib.qavit_tl_bwd_parts(B, N, M))
    R = M * Cc + M + 2 * Cc
    parts = torch.empty(n, R, dtype=torch.float32, device=x.device)
    a = L.TlBwdArgs()
    a.dxc, a.x, a.p, a.mean, a.rstd = dxc.data_ptr(), x.data_ptr(), p.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    a.ln_g, a.ln_b, a.W, a.dx, a.parts = ln_g.data_ptr(), ln_b.data_ptr(), Wc.data_ptr(), dx.data_ptr(), parts.data_ptr()
    a.B, a.N, a.M, a.C = B, N, M, Cc
    L.check(lib.qavit_tl_bwd(C.byref(a), stream()), "tl_bwd")
    base, keep = parts.data_ptr(), (parts, dW, dbias, dgamma, dbeta)
    if dW is not None:
        half = M * Cc // 2
        step = min(half, 2048)
        while half % step or step % 4:
            step -= 4
        for o in range(0, half, step):                       # a reduce descriptor adds the two halves of a 2c-float slice to two destinations (c <= 2048)
            DeferredLN.push_raw(base + 4 * 2 * o, n, step, dW.data_ptr() + 4 * 2 * o, dW.data_ptr() + 4 * (2 * o + step), R, keep)
    if dbias is not None:
        DeferredLN.push_raw(base + 4 * M * Cc, n, M // 2, dbias.data_ptr(), dbias.data_ptr() + 4 * (M // 2), R, keep)
    if dgamma is not None or dbeta is not None:
        DeferredLN.push_raw(base + 4 * (M * Cc + M), n, Cc, _p(dgamma), _p(dbeta), R, keep)


def tokmix_fwd(scores, x, p, xc, B, N, M, Cc):
    L.check(L.load().qavit_tokmix_fwd(dt_code(x.dtype), scores.data_ptr(), x.data_ptr(), p.data_ptr(), xc.data_ptr(), B, N, M, Cc, stream()), "tokmix_fwd")


def tokmix_bwd(p, x, dxc, dx, dscores, B, N, M, Cc):
    L.check(L.load().qavit_tokmix_bwd(dt_code(x.dtype), p.data_ptr(), x.data_ptr(), dxc.data_ptr(), dx.data_ptr(), dscores.data_ptr(), B, N, M, Cc, stream()), "tokmix_bwd")


def upmix_fwd(xc, W, bias, gamma, beta, eps, y, mean, rstd, B, N, M, Cc):
    L.check(L.load().qavit_upmix_fwd(dt_code(xc.dtype), xc.data_ptr(), W.data_ptr(), bias.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps,
                                     y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, N, M, Cc, stream()), "upmix_fwd")


def upmix_fwd_sa_ok(x, u, N, M, Cc) -> bool:
    return (bool(L.load().qavit_upmix_fwd_sa_supported(dt_code(x.dtype), N, M, Cc)) and x.data_ptr() % 8 == 0 and u.data_ptr() % 8 == 0
            and x.dtype == u.dtype)


def upmix_fwd_sa(x, u, sa_gamma, dp, rng, xc, W, bias, gamma, beta, eps, y, mean, rstd, B, N, M, Cc):
    """xc = x + droppath(sa_gamma * u) and y = upmix(xc) in one launch (qavit_upmix_fwd_sa; check upmix_fwd_sa_ok first).
    dp = (dp_p, dp_site, rows_per_sample) with rows_per_sample == M (samples = images)."""
    L.check(L.load().qavit_upmix_fwd_sa(dt_code(x.dtype), x.data_ptr(), u.data_ptr(), _p(sa_gamma), float(dp[0]), int(dp[1]),
                                        _p(rng) if dp[0] > 0.0 else None, xc.data_ptr(), W.data_ptr(), bias.data_ptr(), gamma.data_ptr(),
                                        beta.data_ptr(), eps, y.data_ptr(), mean.data_ptr(), rstd.data_ptr(), B, N, M, Cc, stream()), "upmix_fwd_sa")


def upmix_bwd_sa_ok(xc, N, M, Cc) -> bool:
    return bool(L.load().qavit_upmix_bwd_sa_supported(dt_code(xc.dtype), N, M, Cc)) and xc.data_ptr() % 8 == 0


def upmix_bwd(dy, xc, W, bias, gamma, mean, rstd, dxc, dW, dbias, dgamma, dbeta, B, N, M, Cc, sa=None):
    """``sa`` = (u, du, gamma_sa, dgamma_sa, (dp_p, dp_site, rows_per_sample), rng): also differentiate the scale-add in front of the up-mix
    (qavit_upmix_bwd_sa; check upmix_bwd_sa_ok first)."""
    lib = L.load()
    # inside a backward pass the parameter gradients leave as one partial row per workgroup and join the pass's single reduce launch
    n = lib.qavit_upmix_bwd_parts(dt_code(xc.dtype), B, N, M, Cc) if (DeferredLN.enabled and DeferredLN.ON and (N * M) % 8 == 0 and N % 8 == 0) else 0
    parts = torch.empty(n, N * M + N + 2 * Cc + 4, dtype=torch.float32, device=xc.device) if n > 0 else None     # QAVIT_UPMIX_PART_ROW
    if sa is not None:
        u, du, g_sa, dg_sa, dp, rng = sa
        L.check(lib.qavit_upmix_bwd_sa(dt_code(xc.dtype), dy.data_ptr(), xc.data_ptr(), W.data_ptr(), bias.data_ptr(), gamma.data_ptr(),
                                       mean.data_ptr(), rstd.data_ptr(), dxc.data_ptr(), dW.data_ptr(), _p(dbias), dgamma.data_ptr(), dbeta.data_ptr(),
                                       B, N, M, Cc, _p(parts), u.data_ptr(), du.data_ptr(), _p(g_sa), _p(dg_sa), float(dp[0]), int(dp[1]),
                                       _p(rng) if dp[0] > 0.0 else None, stream()), "upmix_bwd_sa")
    else:
        L.check(lib.qavit_upmix_bwd_p(dt_code(xc.dtype), dy.data_ptr(), xc.data_ptr(), W.data_ptr(), bias.data_ptr(), gamma.data_ptr(),
                                      mean.data_ptr(), rstd.data_ptr(), dxc.data_ptr(), dW.data_ptr(), _p(dbias), dgamma.data_ptr(), dbeta.data_ptr(),
                                      B, N, M, Cc, _p(parts), stream()), "upmix_bwd")
    if parts is None:
        return
    R, base, keep = parts.shape[1], parts.data_ptr(), (parts, dW, dbias, dgamma, dbeta)
    half = N * M // 2                                        # a reduce descriptor adds the two halves of a 2c-float slice to two destinations
    step = min(half, 2048)
    for o in range(0, half, step):                           # dW in slices the reduce kernel takes (c <= 2048)
        DeferredLN.push_raw(base + 4 * 2 * o, n, step, dW.data_ptr() + 4 * 2 * o, dW.data_ptr() + 4 * (2 * o + step), R, keep)
    if dbias is not None:
        DeferredLN.push_raw(base + 4 * N * M, n, N // 2, dbias.data_ptr(), dbias.data_ptr() + 4 * (N // 2), R, keep)
    DeferredLN.push_raw(base + 4 * (N * M + N), n, Cc, dgamma.data_ptr(), dbeta.data_ptr(), R, keep)
    if sa is not None and sa[3] is not None:                 # the layer scale in front of the up-mix: the last four floats of each row
        DeferredLN.push_raw(base + 4 * (N * M + N + 2 * Cc), n, 1, sa[3].data_ptr(), None, R, keep + (sa[3],))


def gather_pool_fwd(x, idx, y, B, N, NP, stride, Cc):
    L.check(L.load().qavit_gather_pool_fwd(dt_code(x.dtype), x.data_ptr(), idx.data_ptr(), y.data_ptr(), B, N, NP, stride, Cc, stream()), "gather_pool_fwd")


def gather_pool_bwd(dy, idx, dx, B, N, NP, stride, Cc):
    L.check(L.load().qavit_gather_pool_bwd(dt_code(dy.dtype), dy.data_ptr(), idx.data_ptr(), dx.data_ptr(), B, N, NP, stride, Cc, stream()), "gather_pool_bwd")


def gather_pool_bwd_ld(dy, idx, dx_ptr, ldx, B, N, NP, stride, Cc):
    """gather_pool_bwd into a column slice: ``dx_ptr`` = address of the slice's first element, rows ``ldx`` elements apart."""
    L.check(L.load().qavit_gather_pool_bwd_ld(dt_code(dy.dtype), dy.data_ptr(), idx.data_ptr(), dx_ptr, ldx, B, N, NP, stride, Cc, stream()), "gather_pool_bwd_ld")


def token_mean_fwd(x, y, B, N, Cc):
    L.check(L.load().qavit_token_mean_fwd(dt_code(x.dtype), x.data_ptr(), y.data_ptr(), B, N, Cc, stream()), "token_mean_fwd")


def token_mean_bwd(dy, dx, B, N, Cc):
    L.check(L.load().qavit_token_mean_bwd(dt_code(dy.dtype), dy.data_ptr(), dx.data_ptr(), B, N, Cc, stream()), "token_mean_bwd")


# ---------------------------------------------------------------------------------------------------
# CCF-FFN middle
# ---------------------------------------------------------------------------------------------------
def ccf_args(dtype, flags, B, Hs, Ws, Cc, eps):
    a = L.CcfArgs()
    a.dtype, a.flags, a.B, a.Hs, a.Ws, a.C, a.eps = dt_code(dtype), flags, B, Hs, Ws, Cc, eps
    return a


def ccf_fwd(a):
    L.check(L.load().qavit_ccf_mid_fwd(C.byref(a), stream()), "ccf_mid_fwd")


def ccf_bwd(a):
    L.check(L.load().qavit_ccf_mid_bwd(C.byref(a), stream()), "ccf_mid_bwd")


def im2col(src, nchw_f32, cols, B, Cin, H, W, k, stride, pad):
    L.check(L.load().qavit_im2col_ld(dt_code(cols.dtype), src.data_ptr(), 1 if nchw_f32 else 0, cols.data_ptr(), cols.shape[1], B, Cin, H, W, k, stride,
                                     pad, stream()), "im2col")


def col2im(dcols, dx, B, Cin, H, W, k, stride, pad):
    L.check(L.load().qavit_col2im(dt_code(dcols.dtype), dcols.data_ptr(), dx.data_ptr(), B, Cin, H, W, k, stride, pad, stream()), "col2im")


def dwconv_fwd(x, w, bias, y, B, H, W, Cc, ks):
    L.check(L.load().qavit_dwconv_fwd(dt_code(x.dtype), x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), B, H, W, Cc, ks, stream()), "dwconv_fwd")


def dwconv_fwd_ld(x, w, bias, y, ldy, B, H, W, Cc, ks):
    """``y`` is a column slice (row stride ``ldy`` elements) of a wider buffer."""
    L.check(L.load().qavit_dwconv_fwd_ld(dt_code(x.dtype), x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), ldy, B, H, W, Cc, ks, stream()), "dwconv_fwd_ld")


def dwconv_fwd_ld2(x, w, bias, y, ldy, xcopy, ldc, B, H, W, Cc, ks):
    """As dwconv_fwd_ld, and the same launch copies ``x`` into the column slice ``xcopy`` (row stride ``ldc``): the pass-through member of
    LMFAdapter's cat.  Map sides multiples of 8."""
    L.check(L.load().qavit_dwconv_fwd_ld2(dt_code(x.dtype), x.data_ptr(), w.data_ptr(), _p(bias), y.data_ptr(), ldy, xcopy.data_ptr(), ldc,
                                          B, H, W, Cc, ks, stream()), "dwconv_fwd_ld2")


def dwconv_bwd_ld(dy, lddy, x, w, dx, dadd, lddadd, dw, dbias, B, H, W, Cc, ks):
    """``dy`` is a column slice (row stride ``lddy``); ``dadd`` (None, or row stride ``lddadd``; may be ``dx``) is added into ``dx``."""
    L.check(L.load().qavit_dwconv_bwd_ld(dt_code(x.dtype), dy.data_ptr(), lddy, x.data_ptr(), w.data_ptr(), dx.data_ptr(), _p(dadd), lddadd,
                                         dw.data_ptr(), _p(dbias), B, H, W, Cc, ks, stream()), "dwconv_bwd_ld")


def dwconv_bwd(dy, x, w, dx, dw, dbias, B, H, W, Cc, ks):
    L.check(L.load().qavit_dwconv_bwd(dt_code(x.dtype), dy.data_ptr(), x.data_ptr(), w.data_ptr(), dx.data_ptr(), dw.data_ptr(), _p(dbias),
                                      B, H, W, Cc, ks, stream()), "dwconv_bwd")


# ---------------------------------------------------------------------------------------------------
# bank
# ---------------------------------------------------------------------------------------------------
def bank_stats(tokens, g_branch, b_branch, g_write, b_write, Wg, bg, acc, ws, B, N, Cc, S, eps, fix=None):
    """``fix`` (lib.NanFix): the NaN -> zeros rule a fused branch call deferred to this launch (qavit_bank_stats_nanfix)."""
    L.check(L.load().qavit_bank_stats_nanfix(dt_code(tokens.dtype), tokens.data_ptr(), g_branch.data_ptr(), b_branch.data_ptr(), g_write.data_ptr(),
                                             b_write.data_ptr(), Wg.data_ptr(), bg.data_ptr(), _p(acc), ws.data_ptr(), ws.numel(),
                                             B, N, Cc, S, eps, C.byref(fix) if fix is not None else None, stream()), "bank_stats")


def bank_ws_floats(B, N, Cc, S) -> int:
    return int(L.load().qavit_bank_ws_floats(B, N, Cc, S))


def bank_apply(acc, Wc, bc, bank_k, bank_v, update_count, S, Cc, inv_batch, mode, parts=None, nparts=0, snap=None):
    L.check(L.load().qavit_bank_apply(acc.data_ptr(), Wc.data_ptr(), bc.data_ptr(), bank_k.data_ptr(), bank_v.data_ptr(), _p(update_count),
                                      S, Cc, inv_batch, mode, _p(parts), int(nparts), _p(snap[0]) if snap else None, _p(snap[1]) if snap else None,
                                      stream()), "bank_apply")


# ---------------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------------
def patchify(img, cols, B, Cin, H, W, p):
    L.check(L.load().qavit_patchify(dt_code(cols.dtype), img.data_ptr(), cols.data_ptr(), B, Cin, H, W, p, stream()), "patchify")


def hybrid_fuse_fwd(x, fw, y, rows, nb, Cb):
    L.check(L.load().qavit_hybrid_fuse_fwd(dt_code(x.dtype), x.data_ptr(), fw.data_ptr(), y.data_ptr(), rows, nb, Cb, stream()), "hybrid_fuse_fwd")


NARROW_PARTS_MAX = 1024                                     # QAVIT_NARROW_PARTS_MAX


def _narrow_ws(device, want, row=4):
    """Workspace for the narrow partial rows of a 1-4 value parameter gradient (include/qavit.h, NARROW PARTIAL ROWS), or None: float atomics.
    Every workgroup adding its share to ONE address made such a gradient differ in its last bit from run to run."""
    if not (want and DeferredLN.ON):
        return None
    return torch.empty(NARROW_PARTS_MAX * row, dtype=torch.float32, device=device)


def _narrow_push(ws, n, Cn, dst, row=4):
    """Fold the n rows a launch left in ``ws`` into ``dst`` (fixed order): in the pass's one reduce launch, or now."""
    if ws is not None and n.value > 0:
        DeferredLN.push_raw(ws.data_ptr(), int(n.value), Cn, dst.data_ptr(), None, row, (ws, dst))


def hybrid_fuse_bwd(dy, x, fw, dx, dfw, rows, nb, Cb):
    ws, n = _narrow_ws(x.device, nb <= 4, 8), C.c_int(0)
    L.check(L.load().qavit_hybrid_fuse_bwd(dt_code(x.dtype), dy.data_ptr(), x.data_ptr(), fw.data_ptr(), dx.data_ptr(), dfw.data_ptr(), rows, nb, Cb,
                                           _p(ws), C.byref(n), stream()), "hybrid_fuse_bwd")
    _narrow_push(ws, n, 4, dfw, 8)


def sum_k(xs, out):
    L.check(L.load().qavit_sum_k(dt_code(out.dtype), _ptr_arr(xs), len(xs), out.data_ptr(), out.numel(), stream()), "sum_k")


def rand_perm(perm, B, rng, site):
    L.check(L.load().qavit_rand_perm(perm.data_ptr(), B, rng.data_ptr(), site, stream()), "rand_perm")


def mix_apply(x, perm, plan, out):
    B, Cc, H, W = x.shape
    L.check(L.load().qavit_mix_apply(x.data_ptr(), perm.data_ptr(), plan.data_ptr(), out.data_ptr(), B, Cc, H, W, stream()), "mix_apply")


def gate_mix_fwd(t, r, g, y):
    L.check(L.load().qavit_gate_mix_fwd(dt_code(t.dtype), t.data_ptr(), r.data_ptr(), g.data_ptr(), y.data_ptr(), t.numel(), stream()), "gate_mix_fwd")


def gate_mix_bwd(dy, r, g, dr, dg):
    L.check(L.load().qavit_gate_mix_bwd(dt_code(r.dtype), dy.data_ptr(), r.data_ptr(), g.data_ptr(), dr.data_ptr(), dg.data_ptr(), r.numel(), stream()), "gate_mix_bwd")


def mix2_fwd(a, b, fw, y):
    L.check(L.load().qavit_mix2_fwd(dt_code(a.dtype), a.data_ptr(), b.data_ptr(), fw.data_ptr(), y.data_ptr(), a.numel(), stream()), "mix2_fwd")


def mix3_fwd(a, t, h, fw, y, drop, rng):
    L.check(L.load().qavit_mix3_fwd(dt_code(a.dtype), a.data_ptr(), t.data_ptr(), h.data_ptr(), fw.data_ptr(), y.data_ptr(), a.numel(),
                                    drop[0], drop[1], rng.data_ptr(), stream()), "mix3_fwd")


def mix3_bwd(dy, a, t, h, fw, da, dt, dh, dfw, drop, rng):
    ws, n = _narrow_ws(a.device, dfw is not None), C.c_int(0)
    L.check(L.load().qavit_mix3_bwd(dt_code(a.dtype), dy.data_ptr(), a.data_ptr(), t.data_ptr(), h.data_ptr(), fw.data_ptr(), da.data_ptr(), dt.data_ptr(),
                                    dh.data_ptr(), _p(dfw), a.numel(), drop[0], drop[1], rng.data_ptr(), _p(ws), C.byref(n), stream()), "mix3_bwd")
    _narrow_push(ws, n, 2, dfw)


def mix3_ln_ok(a, t, h, Cc) -> bool:
    vec = 4 * a.element_size()
    return (bool(L.load().qavit_mix3_ln_supported(dt_code(a.dtype), Cc)) if a.dtype in (torch.bfloat16, torch.float32) else False) and \
        a.dtype == t.dtype == h.dtype and a.numel() == t.numel() == h.numel() and a.numel() < 2 ** 32 and \
        all(u.is_contiguous() and u.data_ptr() % vec == 0 for u in (a, t, h))


def mix3_ln_fwd(a, t, h, fw, drop, rng, mixed, gamma, beta, eps, y, mean, rstd, rows, Cc):
    """mixed = softmax(fw)[0]*a + softmax(fw)[1]*(t + dropout(h)); y = LayerNorm(mixed): one launch (qavit_mix3_ln_fwd; check mix3_ln_ok)."""
    L.check(L.load().qavit_mix3_ln_fwd(dt_code(a.dtype), a.data_ptr(), t.data_ptr(), h.data_ptr(), fw.data_ptr(), float(drop[0]), int(drop[1]),
                                       _p(rng) if drop[0] > 0.0 else None, mixed.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, y.data_ptr(),
                                       mean.data_ptr(), rstd.data_ptr(), rows, Cc, stream()), "mix3_ln_fwd")


def _mix3_ln_parts(mixed, rows, Cc, want):
    """Partial-row workspace of the fused blend + norm backward (its own workgroup count: n LayerNorm rows of 2C floats, then n narrow
    rows of 4 floats for the blend logits), or None: atomics."""
    if not (want and DeferredLN.enabled and DeferredLN.ON):
        return None
    n = L.load().qavit_mix3_ln_bwd_parts(rows, Cc)
    return torch.empty(n * (2 * Cc + 4), dtype=torch.float32, device=mixed.device), n


def _mix3_ln_push(part, Cc, dfw, dgamma, dbeta):
    ws, n = part
    DeferredLN.push(ws, n, Cc, dgamma, dbeta)
    if dfw is not None:
        DeferredLN.push_raw(ws.data_ptr() + 4 * n * 2 * Cc, n, 2, dfw.data_ptr(), None, 4, (ws, dfw))


def mix3_ln_bwd(dy, a, t, h, fw, drop, rng, mixed, gamma, mean, rstd, da, dt, dh, dfw, dgamma, dbeta, rows, Cc):
    """LayerNorm backward + blend backward in one launch (qavit_mix3_ln_bwd); the LayerNorm parameter gradients as partial rows when a
    backward pass has armed DeferredLN."""
    part = _mix3_ln_parts(mixed, rows, Cc, dgamma is not None or dbeta is not None)
    L.check(L.load().qavit_mix3_ln_bwd(dt_code(a.dtype), dy.data_ptr(), a.data_ptr(), t.data_ptr(), h.data_ptr(), fw.data_ptr(), float(drop[0]), int(drop[1]),
                                       _p(rng) if drop[0] > 0.0 else None, mixed.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                       da.data_ptr(), dt.data_ptr(), dh.data_ptr(), _p(dfw), _p(dgamma), _p(dbeta), rows, Cc,
                                       part[0].data_ptr() if part else None, stream()), "mix3_ln_bwd")
    if part:
        _mix3_ln_push(part, Cc, dfw, dgamma, dbeta)


def gate_mix3_ln_fwd(t, r, g, h, fw, drop, rng, mixed, gamma, beta, eps, y, mean, rstd, rows, Cc):
    """y = LayerNorm(s0*(t + sigmoid(g)*r) + s1*(t + dropout(h))): SplitFusion's gate, blend and final norm in one launch
    (qavit_gate_mix3_ln_fwd; operands as for mix3_ln_ok)."""
    L.check(L.load().qavit_gate_mix3_ln_fwd(dt_code(t.dtype), t.data_ptr(), r.data_ptr(), g.data_ptr(), h.data_ptr(), fw.data_ptr(), float(drop[0]), int(drop[1]),
                                            _p(rng) if drop[0] > 0.0 else None, mixed.data_ptr(), gamma.data_ptr(), beta.data_ptr(), eps, y.data_ptr(),
                                            mean.data_ptr(), rstd.data_ptr(), rows, Cc, stream()), "gate_mix3_ln_fwd")


def gate_mix3_ln_bwd(dy, t, r, g, h, fw, drop, rng, mixed, gamma, mean, rstd, dt, dr, dg, dh, dfw, dgamma, dbeta, rows, Cc):
    part = _mix3_ln_parts(mixed, rows, Cc, dgamma is not None or dbeta is not None)
    L.check(L.load().qavit_gate_mix3_ln_bwd(dt_code(t.dtype), dy.data_ptr(), t.data_ptr(), r.data_ptr(), g.data_ptr(), h.data_ptr(), fw.data_ptr(),
                                            float(drop[0]), int(drop[1]), _p(rng) if drop[0] > 0.0 else None, mixed.data_ptr(), gamma.data_ptr(),
                                            mean.data_ptr(), rstd.data_ptr(), dt.data_ptr(), dr.data_ptr(), dg.data_ptr(), dh.data_ptr(), _p(dfw),
                                            _p(dgamma), _p(dbeta), rows, Cc, part[0].data_ptr() if part else None, stream()), "gate_mix3_ln_bwd")
    if part:
        _mix3_ln_push(part, Cc, dfw, dgamma, dbeta)


def mix2_bwd(dy, a, b, fw, da, db, dfw):
    ws, n = _narrow_ws(a.device, dfw is not None), C.c_int(0)
    L.check(L.load().qavit_mix2_bwd(dt_code(a.dtype), dy.data_ptr(), a.data_ptr(), b.data_ptr(), fw.data_ptr(), da.data_ptr(), db.data_ptr(),
                                    _p(dfw), a.numel(), _p(ws), C.byref(n), stream()), "mix2_bwd")
    _narrow_push(ws, n, 2, dfw)


def scale_add_fwd(x, u, gamma, y, rows, Cc, dp, rng):
    L.check(L.load().qavit_scale_add_fwd(dt_code(x.dtype), x.data_ptr(), u.data_ptr(), _p(gamma), y.data_ptr(), rows, Cc, dp[0], dp[1], dp[2], _p(rng), stream()), "scale_add_fwd")


def scale_add_bwd(dy, u, gamma, du, dgamma, rows, Cc, dp, rng):
    ws, n = _narrow_ws(dy.device, dgamma is not None), C.c_int(0)
    L.check(L.load().qavit_scale_add_bwd(dt_code(dy.dtype), dy.data_ptr(), u.data_ptr(), _p(gamma), du.data_ptr(), _p(dgamma), rows, Cc, dp[0], dp[1], dp[2], _p(rng),
                                         _p(ws), C.byref(n), stream()), "scale_add_bwd")
    _narrow_push(ws, n, 1, dgamma)


def chan_scale_add_fwd(x, u, gamma, y, rows, Cc, dp, rng):
    L.check(L.load().qavit_chan_scale_add_fwd(dt_code(x.dtype), x.data_ptr(), u.data_ptr(), gamma.data_ptr(), y.data_ptr(), rows, Cc,
                                              dp[0], dp[1], dp[2], _p(rng), stream()), "chan_scale_add_fwd")


def chan_scale_add_bwd(dy, u, gamma, du, dgamma, rows, Cc, dp, rng):
    L.check(L.load().qavit_chan_scale_add_bwd(dt_code(dy.dtype), dy.data_ptr(), u.data_ptr(), gamma.data_ptr(), du.data_ptr(), dgamma.data_ptr(),
                                              rows, Cc, dp[0], dp[1], dp[2], _p(rng), stream()), "chan_scale_add_bwd")


def bn_supported(dtype, Cc: int) -> bool:
    vec = 8 if dtype == torch.bfloat16 else 4
    return Cc % vec == 0 and Cc // vec <= 256 and 256 % (Cc // vec) == 0 and Cc <= 2048


def bn_fwd(x, y, M, Cc, gamma, beta, running_mean, running_var, momentum, eps, act, save_mean, save_rstd, ws, training, phase=0, m_total=0):
    L.check(L.load().qavit_bn_fwd(dt_code(x.dtype), x.data_ptr(), y.data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(),
                                  _p(running_mean), _p(running_var), float(momentum), float(eps), int(act),
                                  _p(save_mean), _p(save_rstd), _p(ws), int(training), int(phase), int(m_total), stream()), "bn_fwd")


def bn_bwd(dy, x, M, Cc, gamma, beta, save_mean, save_rstd, act, training, dx, dgamma, dbeta, ws, phase=0, m_total=0, ws_param=None):
    L.check(L.load().qavit_bn_bwd(dt_code(x.dtype), dy.data_ptr(), x.data_ptr(), M, Cc, gamma.data_ptr(), beta.data_ptr(),
                                  save_mean.data_ptr(), save_rstd.data_ptr(), int(act), int(training), dx.data_ptr(), _p(dgamma), _p(dbeta),
                                  ws.data_ptr(), int(phase), int(m_total), _p(ws_param), stream()), "bn_bwd")


def spatial_ln_supported(N, Cc):
    return Cc % 4 == 0 and N * Cc in (4096, 8192, 16384)


def spatial_ln_fwd(x, w, b, y, mean, rstd, B, N, Cc, eps):
    L.check(L.load().qavit_spatial_ln_fwd(dt_code(x.dtype), x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), mean.data_ptr(),
                                          rstd.data_ptr(), B, N, Cc, float(eps), stream()), "spatial_ln_fwd")


def spatial_ln_bwd(dy, x, w, mean, rstd, dx, dw, db, B, N, Cc):
    L.check(L.load().qavit_spatial_ln_bwd(dt_code(x.dtype), dy.data_ptr(), x.data_ptr(), w.data_ptr(), mean.data_ptr(), rstd.data_ptr(),
                                          dx.data_ptr(), dw.data_ptr(), db.data_ptr(), B, N, Cc, stream()), "spatial_ln_bwd")


def dropout(x, y, p, site, rng):
    L.check(L.load().qavit_dropout(dt_code(x.dtype), x.data_ptr(), y.data_ptr(), x.numel(), p, site, rng.data_ptr(), stream()), "dropout")


def pack_weights(dtype, descs_dev, n_desc, max_elems):
    L.check(L.load().qavit_pack_weights(dt_code(dtype), descs_dev.data_ptr(), n_desc, max_elems, stream()), "pack_weights")


def adamw(p, g, m, v, skip, lr_dev, b1, b2, eps, wd, step_dev, gnorm_dev, max_norm):
    L.check(L.load().qavit_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _p(skip), p.numel(), lr_dev.data_ptr(), b1, b2, eps, wd,
                                 step_dev.data_ptr(), _p(gnorm_dev), max_norm, stream()), "adamw")


def local_clip(flat_g, seg, clip, ws):
    """seg: int64 device tensor [n, 2] of (offset, length) into flat_g; each segment is clipped to L2 norm ``clip``.
    ws: zero-initialised fp32 [2*n] scratch (left zero by the call)."""
    L.check(L.load().qavit_local_clip(flat_g.data_ptr(), seg.data_ptr(), seg.shape[0], float(clip), ws.data_ptr(), stream()), "local_clip")


def copy2(a, b):
    """-> (a.clone(), b.clone()) for two same-size fp32 tensors in ONE kernel (no memcpy nodes in the step graph)."""
    a, b = a.detach(), b.detach()
    n = a.numel()
    if a.dtype != torch.float32 or b.dtype != torch.float32 or b.numel() != n or n % 4 or not (a.is_contiguous() and b.is_contiguous()) \
            or (a.data_ptr() | b.data_ptr()) & 15:
        return a.clone(), b.clone()
    out = torch.empty(2, n, dtype=torch.float32, device=a.device)
    L.check(L.load().qavit_copy2(a.data_ptr(), b.data_ptr(), out[0].data_ptr(), out[1].data_ptr(), n, stream()), "copy2")
    return out[0].view(a.shape), out[1].view(b.shape)


def ce_label_smooth(logits, y_a, y_b, lam_dev, ls, loss, dlogits):
    B, Cc = logits.shape
    ws = Runtime.get(logits.device).workspace("ce_ticket", 1 + (B + 15) // 16 + 4096, zero=True)     # ticket + per-workgroup partial losses
    L.check(L.load().qavit_ce_label_smooth(dt_code(logits.dtype), logits.data_ptr(), y_a.data_ptr(), _p(y_b), _p(lam_dev), float(ls), B, Cc,
                                           loss.data_ptr(), _p(dlogits), ws.data_ptr(), stream()), "ce_label_smooth")


def l2norm(g, partial, out):
    L.check(L.load().qavit_l2norm(g.data_ptr(), g.numel(), partial.data_ptr(), out.data_ptr(), stream()), "l2norm")
