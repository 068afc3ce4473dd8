"""End-to-end sanity of the captured bf16 training step: memorise a small fixed synthetic set (random labels).
Loss must fall and every parameter stay finite.  usage: python tools/train_sanity.py [steps] [v1|v2|tin]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
stem = sys.argv[2] if len(sys.argv) > 2 else "v1"
torch.manual_seed(0)
tin = stem == "tin"
cfg = Q.HQAViTTinyINConfig() if tin else Q.HQAViTConfig()
model = Q.HQAViT(cfg, stem="v1" if tin else stem).cuda().train()
B, NSET = (256, 1024) if tin else (1024, 4096)
g = torch.Generator().manual_seed(1)
X = torch.randn(NSET, 3, cfg.img_size, cfg.img_size, generator=g).cuda()
Y = torch.randint(0, cfg.num_classes, (NSET,), generator=g).cuda()
tcfg = Q.TrainingConfig(batch_size=B, use_amp=True)
tr = Q.Trainer(model, tcfg, total_steps=steps, warmup_steps=max(steps // 10, 1), compute_dtype=torch.bfloat16)
tr.capture(X[:B], Y[:B], with_optim=True, warmup=2)
t0 = time.time()
hist = []
for s in range(steps):
    i = (s * B) % NSET
    loss = tr.replay(X[i:i + B], Y[i:i + B])
    if s % 25 == 0 or s == steps - 1:
        hist.append((s, float(loss)))
torch.cuda.synchronize()
print(f"stem={stem} {steps} steps in {time.time() - t0:.1f} s")
print(" ".join(f"{s}:{l:.3f}" for s, l in hist))
bad = [n for n, p in zip(tr.names, tr.params) if not torch.isfinite(p).all()]
print("non-finite parameters:", len(bad), "| bank update_count:", int(model.global_bank.update_count))
assert not bad and hist[-1][1] < hist[0][1] - 0.5, "loss did not fall"
print("OK")
