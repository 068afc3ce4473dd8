import sys, numpy as np, torch
sys.path.insert(0, '.')
import qavit_amd as Q
g = np.load('tests/golden/golden_v1.npz')
x = torch.from_numpy(g['c100/x']).cuda()
def build():
    m = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(m); return m.cuda().eval()
a = build()
outs = {}
def hook(name):
    def f(m, i, o): outs.setdefault(name, []).append(o.detach().float().clone() if torch.is_tensor(o) else None)
    return f
for n, m in a.named_modules():
    if n and n.count('.') <= 3: m.register_forward_hook(hook(n))
with torch.no_grad():
    y1 = a(x); y2 = a(x)
print('same-model repeat diff', (y1-y2).abs().max().item())
first = None
for n, v in outs.items():
    if len(v) == 2 and v[0] is not None:
        d = (v[0]-v[1]).abs().max().item()
        if d > 0 and first is None:
            first = n; print('first nondeterministic module:', n, d)
b = Q.HQAViT(Q.HQAViTConfig()).cuda().eval(); b.load_state_dict(a.state_dict(), strict=True)
with torch.no_grad():
    yb = b(x)
print('a vs b', (y1-yb).abs().max().item(), 'logit scale', y1.abs().max().item())
