"""Device time of the fused channel-group branch (csrc/cga.hip, cga64.hip), forward and forward + backward, hipGraph-replayed.
usage: python tools/bench_cga.py [B] [T] [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import qavit_amd as Q
F = import_module("qa-vit_amd.functional"); M = import_module("qa-vit_amd.modules")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
T = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
cfg = Q.HQAViTConfig()
bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).cuda()
rt = M._Ctx("hqa"); rt.bank_writes = False
mod = M.EfficientChannelGroupAttention(cfg, bank, rt).cuda().train()
x = torch.randn(B, T, cfg.embed_dim, device="cuda").to(torch.bfloat16).requires_grad_(True)
g = torch.randn(B, T, cfg.embed_dim, device="cuda").to(torch.bfloat16)
def fwd():
    with torch.no_grad(): return mod(x)
def both():
    out = mod(x); out.backward(g); x.grad = None
for name, fn, n in (("forward", fwd, 10), ("forward + backward", both, 5)):
    s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(3): fn()
    torch.cuda.current_stream().wait_stream(s_); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(n): fn()
    gph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gph.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"cga B={B} T={T} {name:20s}: {e0.elapsed_time(e1) * 1e3 / (reps * n):8.2f} us", flush=True)
