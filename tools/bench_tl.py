"""Device time of the fused TokenLearner launches (qavit_tl_fwd / qavit_tl_bwd) against the chain they replace, B images of 64 tokens.
usage: bench_tl.py [B] [iters]   (QAVIT_TL_FWD_GRID / QAVIT_TL_BWD_GRID select the workgroup counts)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import importlib  # noqa: E402
import qavit_amd as Q  # noqa: E402

Q.lib.load()
F = importlib.import_module("qa-vit_amd.functional")
M_ = importlib.import_module("qa-vit_amd.modules")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
N, M, C = 64, 16, 192
tl = M_.TokenLearner(C, M).cuda()
x0 = torch.randn(B, N, C, device="cuda").to(torch.bfloat16)
go = torch.randn(B, M, C, device="cuda").to(torch.bfloat16)


def run(fused):
    F._TL_FUSED = fused
    x = x0.clone().requires_grad_(True)
    ts = {"fwd": [], "bwd": []}
    for it in range(iters + 5):
        for p in tl.parameters():
            p.grad = None
        x.grad = None
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        torch.cuda._sleep(200000)
        e[0].record()
        xc = tl(x)
        e[1].record()
        torch.cuda._sleep(200000)
        e[2].record()
        xc.backward(go)
        e[3].record()
        torch.cuda.synchronize()
        if it >= 5:
            ts["fwd"].append(e[0].elapsed_time(e[1]) * 1e3)
            ts["bwd"].append(e[2].elapsed_time(e[3]) * 1e3)
    F._TL_FUSED = True
    return {k: sorted(v)[len(v) // 2] for k, v in ts.items()}


for fused in (True, False):
    r = run(fused)
    print(f"B={B} {'fused  ' if fused else 'unfused'} forward {r['fwd']:7.1f} us   backward {r['bwd']:7.1f} us (eager, incl. the weight-gradient / reduce launches it triggers)")
