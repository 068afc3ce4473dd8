import sys, os, time, torch, importlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels"); par = importlib.import_module("qa-vit_amd.parallel")
Q.lib.load()
torch.cuda.set_device(0); dev = torch.device("cuda", 0)
cfg = Q.HQAViTConfig(); model = Q.HQAViT(cfg); Q.fill_module(model); model = model.to(dev).train()
B = 1024
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).to(dev); y = torch.randint(0, cfg.num_classes, (B,), generator=g).to(dev)
tcfg = Q.TrainingConfig(batch_size=B, use_amp=True)
tr = Q.Trainer(model, tcfg, total_steps=100000, warmup_steps=1000, reducer=None, compute_dtype=torch.bfloat16, order=par.bucket_order)
tr.capture(x, y, with_optim=True, warmup=3)
every = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for i in range(80):
    tr.replay()
    if every and (i + 1) % every == 0:
        torch.cuda.synchronize()
        gn = tr.grad_norm()
        print(i + 1, float(tr.loss), gn, flush=True)
        if gn != gn or gn > 1e3:
            offenders = []
            for n, p in model.named_parameters():
                if p.grad is None: continue
                m = p.grad.abs().max()
                m = float(m)
                if m != m or m > 1e2: offenders.append((n, m, tuple(p.shape)))
            print("offenders:", offenders[:12], len(offenders), flush=True)
            break
torch.cuda.synchronize()
print("final", float(tr.loss.item()), tr.grad_norm())
