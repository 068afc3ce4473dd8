"""Microbench gemm_nt variants at the step's fat shapes; checks results against torch (fp32 of bf16 inputs)."""
import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
F = importlib.import_module("qa-vit_amd.functional"); K = importlib.import_module("qa-vit_amd.kernels")
Q.lib.load(); dt = torch.bfloat16; dev = "cuda"
rt = K.Runtime.get(torch.device("cuda:0"))
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
shapes = [(65536, 256, 1024, "am2"), (65536, 1024, 256, "ln_act"), (65536, 256, 1024, "R"), (65536, 1024, 256, ""), (65536, 128, 512, "am2"),
          (65536, 192, 768, ""), (65536, 192, 384, ""), (65536, 384, 192, ""), (65536, 64, 256, "am2"), (65536, 256, 64, "ln_act"),
          (16384, 192, 576, ""), (16384, 576, 192, ""), (65536, 192, 192, ""), (16384, 192, 192, ""), (16384, 192, 192, "am2"), (16384, 96, 192, "ln_act"), (10240, 384, 192, "")]
for (M, N, Kd, mode) in shapes:
    x = (torch.randn(M, Kd, device=dev) * 0.5).to(dt); w = torch.randn(N, Kd, device=dev) * 0.05
    wb = w.to(dt).contiguous(); y = torch.empty(M, N, device=dev, dtype=dt); bias = torch.randn(N, device=dev) * 0.1
    kw = {}
    ref = None
    if mode == "am2":
        Z = torch.randn(M, Kd, device=dev).to(dt); out = torch.empty(M, Kd, device=dev, dtype=dt)
        kw = dict(a_mode=2, bwd=dict(Z=Z, ldz=Kd, act=1, drop=(0.0, 0), dp=(0.0, 0, 1), out=out, ldo=Kd), rng=rt.rng)
        bias = None
        z32 = Z.float(); gg = 0.5 * (1 + torch.erf(z32 * 0.7071067811865476)) + z32 * torch.exp(-0.5 * z32 * z32) * 0.3989422804014327
        dz = (x.float() * gg).to(dt)
        ref = dz.float() @ wb.float().t()
    elif mode == "ln_act":
        g_, b_ = torch.randn(Kd, device=dev) * 0.1 + 1, torch.randn(Kd, device=dev) * 0.1
        mean, rstd = torch.empty(M, device=dev), torch.empty(M, device=dev)
        K.row_stats(x, 1e-5, M, Kd, mean, rstd)
        Zo = torch.empty(M, N, device=dev, dtype=dt)
        kw = dict(a_mode=1, ln=(g_, b_, 1e-5), ln_stats=(mean, rstd), Z=Zo, act=1, rng=rt.rng)
        xn = torch.nn.functional.layer_norm(x.float(), (Kd,), g_, b_).to(dt).float()
        ref = torch.nn.functional.gelu(xn @ wb.float().t() + bias)
    elif mode == "R":
        R = torch.randn(M, N, device=dev).to(dt)
        kw = dict(R=R, ldr=N, rng=rt.rng)
        ref = x.float() @ wb.float().t() + bias + R.float()
    else:
        ref = x.float() @ wb.float().t() + bias
    fn = lambda: K.gemm_nt(x, wb, y, M, N, Kd, Kd, Kd, N, bias, **kw)
    t = timeit(fn)
    err = float((y.float() - ref).abs().max() / ref.abs().max())
    extra = ""
    if mode == "am2":
        extra = f" dz_err {float((out.float() - dz.float()).abs().max()):.3g}"
    print(f"M={M:6d} N={N:4d} K={Kd:4d} {mode:7s} {t:8.1f} us  {2.0*M*N*Kd/t/1e6:7.1f} TF/s  relerr {err:.2e}{extra}")
