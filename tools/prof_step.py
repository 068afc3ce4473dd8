"""Run a few eager training steps for rocprofv3 (--kernel-trace --stats)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dt = torch.bfloat16 if (len(sys.argv) < 4 or sys.argv[3] == "bf16") else torch.float32
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=(dt == torch.bfloat16)), total_steps=1000, warmup_steps=10, compute_dtype=dt)
for _ in range(n):
    tr.step(x, y)
torch.cuda.synchronize()
print("done", tr.grad_norm())
