"""The fan node's input-gradient GEMM [B*16, 192] x K = 960 (csrc/gemm_big.hip), plain and with the LayerNorm-backward epilogue, by row-tile
height (QAVIT_BIG_BM32=0 -> 64-row tiles instead of 32).  usage: bench_gemm_k960.py [M=16384]"""
import os, sys, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
Q.lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
N, Kd = 192, 960
dt, dev = torch.bfloat16, "cuda"
A = (torch.randn(M, Kd, device=dev) * 0.5).to(dt)
Wt = (torch.randn(N, 1344, device=dev) * 0.05).to(dt)
x = torch.randn(M, N, device=dev).to(dt)
add0, add1, R = (torch.randn(M, N, device=dev).to(dt) for _ in range(3))
mean, rstd = torch.randn(M, device=dev) * 0.1, torch.rand(M, device=dev) + 0.5
gam = torch.rand(N, device=dev) + 0.5
dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
out = torch.empty(M, N, device=dev, dtype=dt)


def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


plain = lambda: K.gemm_nt(A, Wt, out, M, N, Kd, Kd, 1344, N, None, R=R, ldr=N)
epi = lambda: K.gemm_nt(A, Wt, out, M, N, Kd, Kd, 1344, N, None, R=R, ldr=N,
                        lnbwd=dict(x=x, mean=mean, rstd=rstd, gamma=gam, dgamma=dg, dbeta=db, adds=[add0, add1]))
def epi_parts():                                            # as inside a backward pass: dgamma / dbeta as partial rows, no atomics
    K.DeferredLN.enabled = True
    epi()
    K.DeferredLN.queue.clear()
    K.DeferredLN.enabled = False


for name, fn in (("plain + R", plain), ("LayerNorm-backward epilogue", epi), ("... with partial rows", epi_parts)):
    t = timeit(fn)
    print(f"M={M} N={N} K={Kd} {name:30s} {t:7.1f} us  {2.0 * M * N * Kd / t / 1e6:6.1f} TFLOP/s   (QAVIT_BIG_BM32={os.environ.get('QAVIT_BIG_BM32', 'default')})")
