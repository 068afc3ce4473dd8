"""Per-shape device time of qavit_attn_fwd / qavit_attn_bwd over one eager training step (method of prof_shapes.py).
usage: python tools/prof_attn_shapes.py [c100|tin] [batch]"""
import sys, os, torch, importlib
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
L = importlib.import_module("qa-vit_amd.lib")
lib = L.load()
which = sys.argv[1] if len(sys.argv) > 1 else "c100"
B = int(sys.argv[2]) if len(sys.argv) > 2 else (1024 if which == "c100" else 512)
cfg = Q.HQAViTConfig() if which == "c100" else Q.HQAViTTinyINConfig()
model = Q.HQAViT(cfg); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, cfg.img_size, cfg.img_size, generator=g).cuda(); y = torch.randint(0, cfg.num_classes, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(batch_size=B, use_amp=True), total_steps=1000, warmup_steps=10, compute_dtype=torch.bfloat16)
for _ in range(2): tr.step(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(20000); e1.record(); torch.cuda.synchronize()
spin = max(int(30.0 / (e0.elapsed_time(e1) * 1e3 / 20000.0)), 1)
pairs = []
for _ in range(64):
    torch.cuda._sleep(spin); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); b.record(); pairs.append((a, b))
torch.cuda.synchronize()
empty = sorted(a.elapsed_time(b) for a, b in pairs)[32]
rec = []
def wrap(name):
    orig = getattr(lib, name)
    def f(a, st):
        g_ = a._obj
        torch.cuda._sleep(spin)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); r = orig(a, st); e.record()
        rec.append(((name[6:], g_.mode, g_.G, g_.Nq, g_.L, g_.H, g_.D, g_.S, g_.KC), s, e))
        return r
    setattr(lib, name, f)
    return orig
o1, o2 = wrap("qavit_attn_fwd"), wrap("qavit_attn_bwd")
tr.step(x, y)
torch.cuda.synchronize()
lib.qavit_attn_fwd, lib.qavit_attn_bwd = o1, o2
agg = defaultdict(lambda: [0, 0.0])
for k, s, e in rec:
    d = agg[k]; d[0] += 1; d[1] += max(s.elapsed_time(e) - empty, 0.0)
print(f"{which} B={B}: total {sum(v[1] for v in agg.values()):.3f} ms")
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k[0]:9s} mode={k[1]} G={k[2]:6d} Nq={k[3]:3d} L={k[4]:3d} H={k[5]} D={k[6]:2d} S={k[7]} KC={k[8]:2d}  x{n:3d}  {ms:7.3f} ms  {ms/n*1e3:7.1f} us/call")
