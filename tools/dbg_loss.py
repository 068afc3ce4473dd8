import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
B = 1024
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(), total_steps=1000, warmup_steps=10)
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 12):
    tr.step(x, y)
    torch.cuda.synchronize()
    bk = model.global_bank.global_k
    print(i, float(tr.loss), float(tr.gnorm), float(bk.abs().max()), bool(torch.isnan(bk).any()), flush=True)
