"""Which python call sites issue the small aten elementwise ops in one eager step (TorchDispatchMode + traceback)."""
import sys, os, torch, traceback
from collections import Counter
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
from torch.utils._python_dispatch import TorchDispatchMode
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(1024, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (1024,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(), total_steps=1000, warmup_steps=10)
for _ in range(2): tr.step(x, y)
torch.cuda.synchronize()
WATCH = None
sites = Counter()
class M(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name not in ('view','reshape','_unsafe_view','detach','empty','empty_like','empty_strided','as_strided','t','transpose','expand','unsqueeze','squeeze','select','slice','alias','permute','_reshape_alias','split','unbind','narrow','size','stride','is_contiguous'):
            st = [f for f in traceback.extract_stack() if "qa-vit_amd" in f.filename]
            where = f"{os.path.basename(st[-1].filename)}:{st[-1].lineno} {st[-1].name}" if st else "engine/other"
            numel = max([a.numel() for a in args if isinstance(a, torch.Tensor)] + [0])
            sites[(name, where, "big" if numel > 1 << 16 else "small")] += 1
        return func(*args, **(kwargs or {}))
with M():
    tr.step(x, y)
torch.cuda.synchronize()
for (n, w, sz), c in sorted(sites.items(), key=lambda kv: -kv[1])[:90]:
    print(f"{c:5d} {n:14s} {sz:5s} {w}")
