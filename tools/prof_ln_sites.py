"""Which call sites issue qavit_layernorm_bwd in one eager step, with their row counts."""
import sys, os, torch, traceback, importlib
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(1024, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (1024,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=True), total_steps=1000, warmup_steps=10, compute_dtype=torch.bfloat16)
for _ in range(2): tr.step(x, y)
cnt = Counter()
for name in ("layernorm_bwd", "layernorm_bwd_multi"):
    orig = getattr(K, name)
    def wrap(*a, _o=orig, _n=name, **kw):
        st = [f for f in traceback.extract_stack() if "functional.py" in f.filename]
        where = f"{st[-1].name}:{st[-1].lineno}" if st else "?"
        rows = a[0].shape[0] if hasattr(a[0], "shape") else -1
        cnt[(_n, where, rows)] += 1
        return _o(*a, **kw)
    setattr(K, name, wrap)
tr.step(x, y)
torch.cuda.synchronize()
for k, c in sorted(cnt.items(), key=lambda kv: -kv[1]): print(c, k)
