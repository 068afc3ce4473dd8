"""Staged timing of the training step (progress lines flushed so a hang is localised)."""
import sys, time, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q

def log(*a):
    print(*a, flush=True)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dt = torch.bfloat16 if (len(sys.argv) < 3 or sys.argv[2] == "bf16") else torch.float32
cfg = Q.HQAViTConfig()
model = Q.HQAViT(cfg); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=(dt == torch.bfloat16)), total_steps=1000, warmup_steps=10, compute_dtype=dt)
log("built", B, dt)
def timed(name, fn, n=3):
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): r = fn()
    torch.cuda.synchronize(); log(f"{name}: {(time.time()-t0)/n*1e3:.2f} ms"); return r
with torch.no_grad():
    model.eval(); timed("eval fwd", lambda: model(x) if dt == torch.float32 else torch.autocast("cuda", dtype=dt)(model)(x) , 2); model.train()
timed("train fwd only", lambda: model(x), 2)
timed("fwd+bwd eager", lambda: tr.fwd_bwd(x, y), 3)
timed("full step eager", lambda: tr.step(x, y), 3)
log("capturing")
tr.capture(x, y, warmup=2)
log("captured")
timed("graph replay", lambda: tr.replay(), 10)
log("loss", float(tr.loss))
