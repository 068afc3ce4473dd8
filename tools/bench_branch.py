"""Device time of one attention branch forward at the benchmark shape (B images x 16 tokens x 192): the fused kernel
(csrc/branch_fwd.hip) against the unfused chain (qkv GEMM -> attention -> proj GEMM), both replayed from a hipGraph.
usage: python tools/bench_branch.py [B] [reps]     (also the program for `rocprofv3 --pmc ... -- python3 tools/bench_branch.py`)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import qavit_amd as Q
F = import_module("qa-vit_amd.functional"); K = import_module("qa-vit_amd.kernels")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
T, C, H, S, KC = 16, 192, 4, 16, 32
dev = "cuda"
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).to(dev)
x = rnd(B, T, C).to(torch.bfloat16)
wqkv, bqkv, wproj, bproj = rnd(3 * C, C, sc=0.05), rnd(3 * C, sc=0.1), rnd(C, C, sc=0.05), rnd(C, sc=0.1)
wq, bq = rnd(C, C, sc=0.05), rnd(C, sc=0.1)
Ek, Ev, Ek2, Ev2 = rnd(16, KC, sc=0.3), rnd(16, KC, sc=0.3), rnd(128, KC, sc=0.3), rnd(128, KC, sc=0.3)
bk, bv = rnd(S, C, sc=0.5), rnd(S, C, sc=0.5)
t = [y * 4 + xx for d in (1, 2) for y in range(0, 4, d) for xx in range(0, 4, d)]
idx = torch.tensor(t, dtype=torch.int32, device=dev); NP = len(t) // 2
sa, sp = K.new_site(), K.new_site()
P = 0.1
def fused(kind, save=False):
    kw = dict(want_o=save, save=save)
    if kind == 0: return F.branch_forward(0, x, wqkv, bqkv, wproj, bproj, Ek, Ev, bk, bv, None, 0, 16, (P, sa), (P, sp), **kw)
    if kind == 1: return F.branch_forward(1, x, wqkv, bqkv, wproj, bproj, Ek2, Ev2, bk, bv, idx, 2, NP, (P, sa), (P, sp), **kw)
    return F.branch_forward(2, x, wq, bq, wproj, bproj, None, None, bk, bv, None, 0, 0, (P, sa), (P, sp), **kw)
def fused_save(kind): return fused(kind, True)
def unfused(kind):
    if kind == 0:
        qkv = F.linear(x, wqkv, bqkv).reshape(B * T, 3 * C)
        spec = dict(mode=0, G=B, Nq=T, L=T, H=H, D=C // H, KC=KC, S=S, groups_per_b=1, q_rows_per_b=T, k_rows_per_b=T, q_off=0, k_off=C, v_off=2 * C, q_rows=B * T, drop=(P, sa))
        o = F.AttnFn.apply(qkv, None, Ek, Ev, bk, bv, spec)
    elif kind == 1:
        pooled = F.GatherPoolFn.apply(x, idx, 2)
        kv = F.linear(pooled, wqkv, bqkv, rows=(C, 2 * C)).reshape(B * NP, 2 * C)
        q = F.linear(x, wqkv, bqkv, rows=(0, C)).reshape(B * T, C)
        spec = dict(mode=0, G=B, Nq=T, L=NP, H=H, D=C // H, KC=KC, S=S, groups_per_b=1, q_rows_per_b=T, k_rows_per_b=NP, q_off=0, k_off=0, v_off=C, q_rows=B * T, drop=(P, sa))
        o = F.AttnFn.apply(q, kv, Ek2, Ev2, bk, bv, spec)
    else:
        q = F.linear(x, wq, bq).reshape(B * T, C)
        spec = dict(mode=1, G=B, Nq=T, L=0, H=H, D=C // H, S=S, q_off=0, k_off=0, v_off=0, q_rows=B * T, drop=(P, sa))
        o = F.AttnFn.apply(q, None, None, None, bk, bv, spec)
    return F.linear(o.reshape(B, T, C), wproj, bproj, drop=(P, sp))
# algorithmic FLOPs per image (needed work only)
FL = {0: 3.54 + 0.39 + 0.59 + 1.18, 1: 1.18 + 2.21 * NP / 10 + 0.25 + 0.59 + 1.18, 2: 1.18 + 0.20 + 1.18}
with torch.no_grad():
    for kind, name in ((0, "swa"), (1, "msda"), (2, "cross")):
        for label, fn in (("fused", fused), ("fused+sv", fused_save), ("unfused", unfused)):
            for _ in range(3): fn(kind)
            torch.cuda.synchronize()
            gph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gph):
                for _ in range(10): fn(kind)
            gph.replay(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): gph.replay()
            e1.record(); torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (reps * 10)
            tf = FL[kind] * 1e6 * B / (us * 1e-6) / 1e12
            print(f"{name:6s} {label:8s} B={B}: {us:8.2f} us / branch forward   {tf:7.1f} TFLOP/s algorithmic = {100 * tf / 2500:5.2f} % of 2.5 PF", flush=True)

# ---- training form: BranchFn forward (saves) + fused backward (csrc/branch_bwd.hip) + the qkv input-gradient GEMM(s), replayed from a hipGraph
def train_pair(kind):
    xg = x.detach().clone().requires_grad_(True)
    if kind == 2:
        ps = [t.detach().clone().requires_grad_(True) for t in (wq, bq, wproj, bproj)]
        sk0, sv0 = bk.detach().clone().requires_grad_(True), bv.detach().clone().requires_grad_(True)
        args = None
    else:
        ps = [t.detach().clone().requires_grad_(True) for t in (wqkv, bqkv, wproj, bproj, Ek if kind == 0 else Ek2, Ev if kind == 0 else Ev2)]
        gk, gv = bk.detach().clone().reshape(1, S, C).requires_grad_(True), bv.detach().clone().reshape(1, S, C).requires_grad_(True)
        meta = dict(kind=kind, attn_drop=(P, sa), proj_drop=(P, sp))
        if kind == 1:
            meta.update(pool_idx=idx, pool_stride=2, Lk=NP)
        args = (xg, ps[0], ps[1], ps[2], ps[3], ps[4], ps[5], gk, gv, meta)
    gout = torch.randn(B, T, C, device=dev).to(torch.bfloat16)
    def step():
        if kind == 2:       # the shared rows are activations here (projections of the bank): a fresh node per step
            out = F.BranchFn.apply(xg, ps[0], ps[1], ps[2], ps[3], None, None, sk0 * 1.0, sv0 * 1.0, dict(kind=2, attn_drop=(P, sa), proj_drop=(P, sp)))
        else:
            out = F.BranchFn.apply(*args)
        out.backward(gout)
        xg.grad = None
    return step
for kind, name in ((0, "swa"), (1, "msda"), (2, "cross")):
    step = train_pair(kind)
    s_ = torch.cuda.Stream(); s_.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s_):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s_); torch.cuda.synchronize()
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        for _ in range(5): step()
    gph.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): gph.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:6s} fwd+bwd  B={B}: {e0.elapsed_time(e1) * 1e3 / (reps * 5):8.2f} us / branch (forward with saves, fused backward, qkv dX GEMM, grouped dW + reduce)", flush=True)
