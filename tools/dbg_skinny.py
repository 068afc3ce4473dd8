import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
Q.lib.load()
for dt in (torch.float32, torch.bfloat16):
    for (M, N, Kd) in ((16, 192, 192), (16, 16, 192), (16, 192, 16), (5, 192, 192), (16, 100, 64), (16, 192, 32)):
        A = torch.randn(M, Kd, device="cuda").to(dt); W = (torch.randn(N, Kd, device="cuda") * 0.1).to(dt); b = torch.randn(N, device="cuda")
        C = torch.full((M, N), float("nan"), device="cuda", dtype=dt)
        K.gemm_nt(A, W, C, M, N, Kd, Kd, Kd, N, b)
        ref = A.float() @ W.float().t() + b
        print(dt, M, N, Kd, float((C.float() - ref).abs().max()), bool(torch.isnan(C).any()))
