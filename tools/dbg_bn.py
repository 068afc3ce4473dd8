import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
import torch.nn.functional as TF
F = importlib.import_module("qa-vit_amd.functional")
Q.lib.load()
torch.manual_seed(0)
for (M, Cc, gelu) in ((262144, 32, True), (65536, 64, True), (65536, 128, False), (65536, 256, False)):
    x0 = (torch.randn(M, Cc, device="cuda") * 1.3 + 0.7).bfloat16()
    w = torch.rand(Cc, device="cuda") + 0.5; b = torch.randn(Cc, device="cuda") * 0.1
    go = torch.randn(M, Cc, device="cuda").bfloat16()
    xr = x0.float().requires_grad_(True); wr = w.clone().requires_grad_(True); br = b.clone().requires_grad_(True)
    ref = TF.batch_norm(xr, torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda"), wr, br, True, 0.1, 1e-5)
    if gelu: ref = TF.gelu(ref)
    ref.backward(go.float())
    worst = [0, 0, 0, 0]
    for it in range(40):
        x = x0.clone().requires_grad_(True); wp = w.clone().requires_grad_(True); bp = b.clone().requires_grad_(True)
        rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
        y = F.BatchNormFn.apply(x, wp, bp, rm, rv, 0.1, 1e-5, gelu, True)
        y.backward(go)
        e = [float((y.float() - ref).abs().max()), float((x.grad.float() - xr.grad).abs().max() / xr.grad.abs().max()),
             float((wp.grad - wr.grad).abs().max() / wr.grad.abs().max()), float((bp.grad - br.grad).abs().max() / br.grad.abs().max())]
        if any(v != v for v in e): print("NaN at it", it, e)
        worst = [max(a, c) for a, c in zip(worst, e)]
    print(M, Cc, gelu, "worst errs y/dx/dw/db", [round(v, 5) for v in worst])
