"""Where do the step's device-to-device copies, zero fills and stock adds come from?

Runs eager training steps under torch.profiler with Python stacks and prints, per stock op (aten::copy_, aten::zero_, aten::fill_,
aten::add, aten::add_, aten::zeros, aten::clone, aten::contiguous), the innermost frames inside this package that issued it, with
counts per step.  The hipGraph replays exactly these launches, so every line is a graph node the step pays ~2-5 us for."""
import sys, os, collections
import torch
from torch.profiler import profile, ProfilerActivity
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=True), total_steps=1000, warmup_steps=10, compute_dtype=torch.bfloat16)
for _ in range(2):
    tr.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=True) as prof:
    tr.step(x, y)
    torch.cuda.synchronize()
WANT = {"aten::copy_", "aten::zero_", "aten::fill_", "aten::add", "aten::add_", "aten::zeros", "aten::clone", "aten::contiguous", "aten::cat", "aten::sum", "aten::mul"}
cnt = collections.Counter()
for ev in prof.events():
    if ev.name not in WANT:
        continue
    frames = [f for f in (ev.stack or []) if "qa-vit_amd" in f or "qavit_amd" in f]
    where = " <- ".join(f.split("/")[-1] for f in frames[:3]) if frames else "(autograd engine / no package frame)"
    shp = str(ev.input_shapes)[:60] if ev.input_shapes else ""
    cnt[(ev.name, where, shp)] += 1
for (name, where, shp), c in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0])):
    print(f"x{c:3d} {name:18s} {shp:60s} {where}")
