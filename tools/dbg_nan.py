import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
B = 1024
mode = sys.argv[1] if len(sys.argv) > 1 else "graph"
for trial in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
    import importlib; par = importlib.import_module("qa-vit_amd.parallel")
    tr = Q.Trainer(model, Q.TrainingConfig(batch_size=B, use_amp=True), total_steps=100000, warmup_steps=1000, compute_dtype=torch.bfloat16, order=par.bucket_order)
    if mode in ("graph", "nosync"):
        tr.capture(x, y, with_optim=True, warmup=3)
    first = None
    for i in range(130):
        if mode in ("graph", "nosync"): tr.replay()
        else: tr.step(x, y)
        if (mode != "nosync") or i in (34, 59, 79, 99, 129):
            gn = tr.grad_norm(); ls = float(tr.loss)
            if gn != gn or gn == float("inf") or ls != ls:
                print("step", i, "gnorm", gn, "loss", ls, flush=True)
                first = i; break
    if first is None:
        print(trial, "no nan in 130 steps; gnorm", tr.grad_norm(), flush=True); continue
    bad_p = [n for n, p in model.named_parameters() if not torch.isfinite(p).all()]
    bad_g = [n for n, p in model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    bad_b = [n for n, b in model.named_buffers() if b.dtype.is_floating_point and not torch.isfinite(b).all()]
    print(trial, "first non-finite gnorm at step", first, "| bad params", len(bad_p), bad_p[:4], "| bad grads", len(bad_g), bad_g[:6], "| bad buffers", bad_b[:4], flush=True)
