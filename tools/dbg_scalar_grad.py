"""How cancellation-dominated are the scalar-parameter gradients (CCF gamma, RRCV beta)?  |sum dy*u| / sum |dy*u| per
scale_add backward call of one fp32 step, and the same sums with dy, u rounded to bf16."""
import sys, os, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
cfg = Q.HQAViTConfig(dropout=0.0, drop_path=0.0)
model = Q.HQAViT(cfg); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(11)
x = torch.randn(640, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (640,), generator=g).cuda()
orig = K.scale_add_bwd
rows = []
def wrap(dy, u, gamma, du, dgamma, *a):
    p = dy.float() * u.float()
    pb = dy.bfloat16().float() * u.bfloat16().float()
    rows.append((float(p.sum()), float(p.abs().sum()), float(pb.sum())))
    return orig(dy, u, gamma, du, dgamma, *a)
K.scale_add_bwd = wrap
loss = torch.nn.functional.cross_entropy(model(x), y, label_smoothing=0.12); loss.backward(); torch.cuda.synchronize()
for s, a, sb in rows:
    print(f"sum {s:+.4e}   sum|.| {a:.4e}   ratio {abs(s)/a:.2e}   inputs rounded to bf16: {sb:+.4e}  ({sb/s:.3f}x)")
