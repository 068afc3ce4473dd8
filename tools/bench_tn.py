"""Microbench: grouped weight-gradient GEMMs (qavit_gemm_tn_grouped) at the step's shapes."""
import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels"); L = importlib.import_module("qa-vit_amd.lib")
dev = "cuda"; dt = torch.bfloat16
def timeit(fn, n=10):
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
cases = [(16384, 192, 192, 12, False), (16384, 192, 192, 1, False), (16384, 48, 192, 12, True), (98304, 16, 32, 12, False), (16384, 192, 96, 12, False),
         (65536, 1024, 256, 1, True), (65536, 256, 1024, 1, False), (65536, 192, 192, 3, True), (16384, 576, 192, 8, False), (65536, 16, 192, 8, True),
         (65536, 192, 384, 4, False), (10240, 384, 192, 8, False), (65536, 256, 64, 4, True), (65536, 64, 256, 4, False)]
for (M, N, Kd, cnt, ln) in cases:
    probs = []
    for i in range(cnt):
        A = torch.randn(M, N, device=dev).to(dt); B = torch.randn(M, Kd, device=dev).to(dt)
        Cg = torch.zeros(N, Kd, device=dev); cs = torch.zeros(N, device=dev)
        lnarg = None
        if ln:
            mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
            K.row_stats(B, 1e-5, M, Kd, mean, rstd)
            lnarg = (torch.ones(Kd, device=dev), torch.zeros(Kd, device=dev), mean, rstd)
        probs.append((A, B, Cg, cs, lnarg))
    def fn():
        K.DeferredTN.enabled = True
        for (A, B, Cg, cs, lnarg) in probs:
            K.gemm_tn(A, B, Cg, M, N, Kd, N, Kd, Kd, cs, ln=lnarg)
        K.DeferredTN.flush(); K.DeferredTN.enabled = False
    t = timeit(fn)
    A, B, Cg, cs, lnarg = probs[0]
    Cg.zero_(); cs.zero_(); fn(); torch.cuda.synchronize()
    Bf = B.float()
    if ln: Bf = torch.nn.functional.layer_norm(Bf, (Kd,)).to(dt).float()
    ref = A.float().t() @ Bf
    err = float((Cg - ref).abs().max() / ref.abs().max()); cerr = float((cs - A.float().sum(0)).abs().max() / A.float().sum(0).abs().max())
    mb = cnt * M * (N + Kd) * 2 / 1e6
    print(f"M={M:6d} N={N:4d} K={Kd:4d} x{cnt:2d} ln={int(ln)} {t:8.1f} us  {t/cnt:7.1f} us/problem {2.0*cnt*M*N*Kd/t/1e6:7.1f} TF/s {mb/t:6.2f} TB/s  err {err:.1e} {cerr:.1e}")
