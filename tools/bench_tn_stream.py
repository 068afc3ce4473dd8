"""How fast does the one-launch weight-gradient kernel stream when every range is the same class?  Homogeneous problem sets of ~0.8 GB
(far beyond the Infinity Cache), GB/s of operand bytes.  The number to hold against the mixed list of a real step (tools/tn_census.py)."""
import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
dev = "cuda"; dt = torch.bfloat16
def timeit(fn, n=5):
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
cases = [(16384, 192, 192, 64, False), (65536, 192, 192, 16, False), (65536, 192, 192, 16, True), (65536, 256, 256, 12, False), (65536, 64, 64, 48, False),
         (65536, 128, 128, 24, False), (16384, 48, 192, 100, True), (98304, 16, 32, 84, False), (65536, 256, 1024, 5, False), (65536, 1024, 256, 5, False),
         (262144, 192, 192, 4, False)]
for (M, N, Kd, cnt, ln) in cases:
    probs = []
    for i in range(cnt):
        A = torch.randn(M, N, device=dev).to(dt); B = torch.randn(M, Kd, device=dev).to(dt)
        Cg = torch.zeros(N, Kd, device=dev); cs = torch.zeros(N, device=dev)
        lnarg = (torch.ones(Kd, device=dev), torch.zeros(Kd, device=dev), torch.zeros(M, device=dev), torch.ones(M, device=dev)) if ln else None
        probs.append((A, B, Cg, cs, lnarg))
    def fn():
        K.DeferredTN.enabled = True; K.DeferredTN.home_stream = None
        for (A, B, Cg, cs, lnarg) in probs:
            K.gemm_tn(A, B, Cg, M, N, Kd, N, Kd, Kd, cs, ln=lnarg)
        K.DeferredTN.flush(); K.DeferredTN.enabled = False
    t = timeit(fn)
    mb = cnt * M * (N + Kd) * 2 / 1e6
    print(f"M={M:6d} N={N:4d} K={Kd:4d} ln={int(ln)} x{cnt:3d}  {mb:7.1f} MB  {t:8.1f} us  {mb / t * 1e3:7.0f} GB/s", flush=True)
    del probs
