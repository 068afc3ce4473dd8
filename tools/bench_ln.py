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
