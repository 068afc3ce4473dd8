"""Shapes of the big aten.add calls issued by the autograd engine (gradient fan-in) in one eager training step."""
import sys, os, torch
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
from torch.utils._python_dispatch import TorchDispatchMode
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(1024, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (1024,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=True), total_steps=1000, warmup_steps=10, compute_dtype=torch.bfloat16)
for _ in range(2): tr.step(x, y)
torch.cuda.synchronize()
cnt = Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in ("add", "add_", "cat", "clone", "copy_", "zeros", "zeros_like", "fill_", "zero_", "mul", "sum"):
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            if name == "cat": ts = list(args[0])
            shp = tuple(ts[0].shape) if ts else tuple(args[0]) if args else ()
            dt = str(ts[0].dtype).replace("torch.", "") if ts else ""
            cnt[(name, shp, dt)] += 1
        return func(*args, **(kwargs or {}))
with M():
    tr.step(x, y)
torch.cuda.synchronize()
for (n, s, d), c in sorted(cnt.items(), key=lambda kv: -kv[1] * max(1, int(torch.tensor(kv[0][1]).prod()) if kv[0][1] else 1))[:40]:
    print(f"{c:4d} {n:10s} {d:9s} {s}")
