"""Per-call-shape device time of the gemm wrappers over one eager training step (HIP events around each call).
Deferred dW GEMMs are disabled so every gemm_tn is timed where it is issued."""
import sys, os
os.environ["QAVIT_DEFER_DW"] = "0"
import torch, importlib
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(1024, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (1024,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(), total_steps=1000, warmup_steps=10)
for _ in range(2): tr.step(x, y)
torch.cuda.synchronize()
rec = []
def wrap(name, keyf):
    orig = getattr(K, name)
    def w(*a, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); r = orig(*a, **kw); e1.record()
        rec.append((name, keyf(a, kw), e0, e1))
        return r
    setattr(K, name, w)
wrap("gemm_nt", lambda a, kw: (a[3], a[4], a[5], "am%d" % kw.get("a_mode", 0), "ln" if kw.get("ln") else "", "act" if kw.get("act") else "", "R" if kw.get("R") is not None else "", "bwdT" if kw.get("bwd") else ""))
wrap("gemm_tn", lambda a, kw: (a[3], a[4], a[5], "ln" if kw.get("ln") else ""))
wrap("layernorm_bwd", lambda a, kw: (a[8], a[9]))
wrap("layernorm_fwd", lambda a, kw: (a[5], a[6]))
wrap("row_stats", lambda a, kw: (a[2], a[3]))
tr.step(x, y)
torch.cuda.synchronize()
agg = defaultdict(lambda: [0, 0.0])
for n, k, e0, e1 in rec:
    d = agg[(n, k)]; d[0] += 1; d[1] += e0.elapsed_time(e1)
tot = defaultdict(float)
for (n, k), (c, ms) in agg.items(): tot[n] += ms
print({k: round(v, 3) for k, v in tot.items()})
for (n, k), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    fl = ""
    if n.startswith("gemm"):
        fl = f"{2.0 * k[0] * k[1] * k[2] * c / ms / 1e9:7.1f} TF/s"
    print(f"{n:14s} {str(k):58s} x{c:3d} {ms:8.3f} ms  {1e3*ms/c:8.1f} us/call {fl}")
