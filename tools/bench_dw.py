"""Microbench: depthwise conv fwd / bwd on the 8x8 lateral-path maps (B=1024), checked against torch conv2d."""
import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
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
B, H, W = 1024, 8, 8
for ks in (3, 5, 7):
    for Cc in (64, 128, 256):
        x = torch.randn(B, H * W, Cc, device=dev).to(dt); dy = torch.randn_like(x)
        w = torch.randn(Cc, 1, ks, ks, device=dev) * 0.1; bias = torch.randn(Cc, device=dev) * 0.1
        y = torch.empty_like(x); dx = torch.empty_like(x); dw = torch.zeros_like(w); db = torch.zeros_like(bias)
        tf = timeit(lambda: K.dwconv_fwd(x, w, bias, y, B, H, W, Cc, ks))
        tb = timeit(lambda: K.dwconv_bwd(dy, x, w, dx, dw, db, B, H, W, Cc, ks))
        dw.zero_(); db.zero_(); K.dwconv_bwd(dy, x, w, dx, dw, db, B, H, W, Cc, ks); torch.cuda.synchronize()
        xr = x.float().reshape(B, H, W, Cc).permute(0, 3, 1, 2).requires_grad_(True); wr = w.clone().requires_grad_(True); br = bias.clone().requires_grad_(True)
        yr = torch.nn.functional.conv2d(xr, wr, br, padding=ks // 2, groups=Cc)
        yr.backward(dy.float().reshape(B, H, W, Cc).permute(0, 3, 1, 2))
        ef = float((y.float() - yr.permute(0, 2, 3, 1).reshape(B, H * W, Cc)).abs().max() / yr.abs().max())
        ex = float((dx.float() - xr.grad.permute(0, 2, 3, 1).reshape(B, H * W, Cc)).abs().max() / xr.grad.abs().max())
        ew = float((dw - wr.grad).abs().max() / wr.grad.abs().max()); eb = float((db - br.grad).abs().max() / br.grad.abs().max())
        mb = B * H * W * Cc * 2 / 1e6
        print(f"ks={ks} C={Cc:3d}: fwd {tf:6.1f} us ({2*mb/tf:5.2f} TB/s)  bwd {tb:6.1f} us ({3*mb/tb:5.2f} TB/s)  err y {ef:.1e} dx {ex:.1e} dw {ew:.1e} db {eb:.1e}")
