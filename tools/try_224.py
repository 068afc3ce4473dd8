"""QA-ViT at 224x224 / patch 16 (N=196, 7x7 windows): product path vs the CPU oracle, forward and a few gradients."""
import sys, os, torch, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import qavit_amd as Q
import qavit_oracle as O
import importlib
K = importlib.import_module("qa-vit_amd.kernels")
_seen = set()
def _wrap(name):
    orig = getattr(K, name)
    def f(a):
        key = (name, a.mode, a.G, a.Nq, a.L, a.H, a.D, a.S, a.KC)
        if key not in _seen:
            _seen.add(key); print("attn", key, flush=True)
        return orig(a)
    setattr(K, name, f)
_wrap("attn_fwd"); _wrap("attn_bwd")
def max_rel(a, b): return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
for variant in ("v1", "v2"):
    cfg = Q.QAViTConfig(dropout=0.0, drop_path=0.0)
    model = Q.QAViT(cfg, variant)
    Q.fill_module(model)
    P = {k: v.clone() for k, v in model.state_dict().items()}
    names = [n for n, _ in model.named_parameters()]
    for n in names: P[n].requires_grad_(True)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(5, 3, 224, 224, generator=g); y = torch.randint(0, 100, (5,), generator=g)
    ref = O.qavit_forward(P, x, cfg, train=True, variant=variant)
    lr = O.loss_fn(ref, y, 0.1)
    lr.backward()
    model = model.cuda().train()
    out = model(x.cuda())
    print(variant, "logits max-rel", max_rel(out.detach().cpu().numpy(), ref.detach().numpy()))
    import torch.nn.functional as TF
    loss = TF.cross_entropy(out.float(), y.cuda(), label_smoothing=0.1)
    loss.backward()
    print(variant, "loss", float(loss), float(lr))
    worst = []
    for n, p in model.named_parameters():
        if p.grad is None or P[n].grad is None: continue
        worst.append((max_rel(p.grad.cpu().numpy(), P[n].grad.numpy()), n, float(P[n].grad.abs().max())))
    worst.sort(reverse=True)
    print(variant, "worst grads", [w for w in worst if not w[1].endswith("k_proj.bias")][:8])
    print(variant, "k_proj.bias ref magnitudes", [w[2] for w in worst if w[1].endswith("k_proj.bias")][:4])
