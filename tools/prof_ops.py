"""torch profiler view of one eager step: which aten ops / autograd nodes issue the D2D copies and small kernels."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
from torch.profiler import profile, ProfilerActivity
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(1024, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (1024,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(), total_steps=1000, warmup_steps=10)
for _ in range(3): tr.step(x, y)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    tr.step(x, y); torch.cuda.synchronize()
ka = prof.key_averages()
rows = sorted(ka, key=lambda e: -e.count)
for e in rows[:45]:
    print(f"{e.count:5d}  cpu {e.cpu_time_total/1e3:8.2f} ms  dev {e.device_time_total/1e3:8.2f} ms  {e.key[:70]}")
# who issues the copies / fills: group by python call site
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof2:
    tr.step(x, y); torch.cuda.synchronize()
from collections import Counter
for opname in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::mul", "aten::add", "aten::add_", "aten::mul_"):
    cnt = Counter()
    for ev in prof2.events():
        if ev.name == opname:
            st = [s for s in (ev.stack or []) if "qa-vit_amd" in s or "qavit" in s]
            cnt[st[0] if st else "?"] += 1
    print("==", opname, sum(cnt.values()))
    for k, v in cnt.most_common(14):
        print(f"   {v:5d}  {k[-90:]}")
