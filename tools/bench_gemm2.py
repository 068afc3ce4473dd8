import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q, importlib
F = importlib.import_module("qa-vit_amd.functional"); K = importlib.import_module("qa-vit_amd.kernels")
Q.lib.load(); dt = torch.bfloat16
shapes = [(16384, 192, 192), (16384, 576, 192), (65536, 192, 192), (16384, 48, 192), (16384, 96, 192), (65536, 1024, 256)]
for (M, N, Kd) in shapes:
    x = torch.randn(M, Kd, device="cuda").to(dt); w = torch.randn(N, Kd, device="cuda") * 0.05
    Wc, Wt = F.pack_for(x.device).get(w, dt); y = torch.empty(M, N, device="cuda", dtype=dt)
    for _ in range(20): K.gemm_nt(x, Wc, y, M, N, Kd, Kd, Kd, N, None)
torch.cuda.synchronize()
