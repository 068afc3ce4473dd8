"""Micro-benchmark of gemm_nt / gemm_tn on the step's dominant shapes (HIP events, same process)."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q, importlib
F = importlib.import_module("qa-vit_amd.functional"); K = importlib.import_module("qa-vit_amd.kernels")
Q.lib.load()
dt = torch.bfloat16
def timeit(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
shapes = [(16384, 576, 192), (16384, 192, 192), (16384, 48, 192), (16384, 96, 192), (16384, 192, 96), (65536, 16, 192), (65536, 192, 192),
          (65536, 256, 64), (65536, 64, 256), (65536, 1024, 256), (65536, 256, 1024), (98304, 48, 32), (262144, 32, 27), (65536, 64, 288), (65536, 192, 384)]
for (M, N, Kd) in shapes:
    x = torch.randn(M, Kd, device="cuda").to(dt); w = torch.randn(N, Kd, device="cuda") * 0.05
    Wc, Wt = F.pack_for(x.device).get(w, dt)
    y = torch.empty(M, N, device="cuda", dtype=dt)
    us = timeit(lambda: K.gemm_nt(x, Wc, y, M, N, Kd, Kd, Kd, N, None))
    ref = (x.float() @ w.t()); err = ((y.float() - ref).abs().max() / ref.abs().max()).item()
    g = torch.zeros(N, Kd, device="cuda")
    us2 = timeit(lambda: K.gemm_tn(y, x, g, M, N, Kd, N, Kd, Kd, None))
    fl = 2.0 * M * N * Kd
    print(f"M={M:6d} N={N:4d} K={Kd:4d}  nt {us:8.1f} us {fl/us/1e6:7.1f} TF (err {err:.1e}) | tn {us2:8.1f} us {fl/us2/1e6:7.1f} TF", flush=True)
