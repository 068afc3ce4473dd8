"""Microbench: layernorm_bwd / row_stats / layernorm_fwd device time at the step's shapes (graph-replayed, 50 calls)."""
import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
dev = torch.device("cuda:0")
def timeit(fn, n=50):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for rows, C in ((65536, 192), (16384, 192), (65536, 64), (65536, 256)):
    x = torch.randn(rows, C, device=dev).bfloat16(); dy = torch.randn_like(x); dx = torch.empty_like(x); y = torch.empty_like(x)
    gm, bt = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    mean, rstd = torch.empty(rows, device=dev), torch.empty(rows, device=dev)
    dg, db = torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    K.row_stats(x, 1e-5, rows, C, mean, rstd)
    t_b = timeit(lambda: K.layernorm_bwd(dy, x, gm, mean, rstd, dx, dg, db, rows, C))
    t_b0 = timeit(lambda: K.layernorm_bwd(dy, x, gm, mean, rstd, dx, None, None, rows, C))          # no parameter-gradient flush
    t_br = timeit(lambda: K.layernorm_bwd(dy, x, gm, mean, rstd, dx, dg, db, rows, C, dres=y))       # + the residual gradient
    t_s = timeit(lambda: K.row_stats(x, 1e-5, rows, C, mean, rstd))
    t_f = timeit(lambda: K.layernorm_fwd(x, y, gm, bt, 1e-5, rows, C, mean, rstd))
    mb = rows * C * 2 / 1e6
    print(f"rows={rows} C={C}: bwd {t_b:6.1f} us ({3*mb/t_b:6.0f} GB/s; without dgamma/dbeta {t_b0:6.1f} us; with dres {t_br:6.1f} us)  stats {t_s:6.1f} us ({mb/t_s:6.0f} GB/s)  fwd {t_f:6.1f} us ({2*mb/t_f:6.0f} GB/s)")

# SplitFusion's closing pair (blend, then LayerNorm): one launch each way (Mix3LayerNormFn) against the two nodes
F = importlib.import_module("qa-vit_amd.functional")
for rows, C in ((65536, 192), (16384, 192)):
    a, t, h = (torch.randn(rows, C, device=dev).bfloat16().requires_grad_(True) for _ in range(3))
    fw = torch.tensor([0.75, 0.25], device=dev, requires_grad=True)
    gm, bt = torch.ones(C, device=dev, requires_grad=True), torch.zeros(C, device=dev, requires_grad=True)
    for p_ in (fw, gm, bt):
        p_.grad = torch.zeros_like(p_)
    gy = torch.randn(rows, C, device=dev).bfloat16()
    drop = (0.1, 4321)
    fused = lambda: F.Mix3LayerNormFn.apply(a, t, h, fw, drop, gm, bt, 1e-5)
    split = lambda: F.layer_norm(F.Mix3Fn.apply(a, t, h, fw, drop), gm, bt, 1e-5)
    def fb(fn):
        def run():
            a.grad = t.grad = h.grad = None
            K.DeferredLN.enabled = True
            try:
                fn().backward(gy)
                K.DeferredLN.flush()
            finally:
                K.DeferredLN.enabled = False
        return run
    with torch.no_grad():
        tf, ts = timeit(fused), timeit(split)
    tfb, tsb = timeit(fb(fused)), timeit(fb(split))
    print(f"rows={rows} C={C}: blend + norm  fwd one launch {tf:6.1f} us / two {ts:6.1f} us;  fwd+bwd (+ the partial-row reduce) one launch each {tfb:6.1f} us / two each {tsb:6.1f} us")
