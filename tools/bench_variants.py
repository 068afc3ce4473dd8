"""Step time of the widened variants (section 8f N4) under the same Trainer + hipGraph as bench.py: the v2-stem HQA-ViT
at 32 px and QA-ViT v1 / v2 at 224 px, bf16, full training step.  Also checks that back-to-back graph replays stay
finite for them (the memset-node race of DESIGN.md section 6 would show here)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q

CASES = {
    "c100v2": (lambda: Q.HQAViT(Q.HQAViTConfig(), stem="v2"), 32, 1024, 100),
    "q224_v1": (lambda: Q.QAViT(Q.QAViTConfig(), "v1"), 224, 128, 100),
    "q224_v2": (lambda: Q.QAViT(Q.QAViTConfig(), "v2"), 224, 128, 100),
}
for name in ([a for a in sys.argv[1:] if a in CASES] if sys.argv[1:] else list(CASES)):
    build, px, B, ncls = CASES[name]
    torch.manual_seed(0)
    model = build(); Q.fill_module(model); model = model.cuda().train()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, px, px, generator=g).cuda(); y = torch.randint(0, ncls, (B,), generator=g).cuda()
    tr = Q.Trainer(model, Q.TrainingConfig(batch_size=B, use_amp=True), total_steps=100000, warmup_steps=1000, compute_dtype=torch.bfloat16)
    tr.capture(x, y, with_optim=True, warmup=3)
    for _ in range(20): tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); n = 30
    for _ in range(n): tr.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    bad = [nm for nm, p in zip(tr.names, tr.params) if not torch.isfinite(p).all()]
    print(f"{name}: B={B} {dt*1e3:.2f} ms/step  {B/dt:.0f} img/s  loss {float(tr.loss):.4f}  grad-norm {tr.grad_norm():.3f}  non-finite params: {len(bad)}", flush=True)
    del tr, model
    torch.cuda.empty_cache()

# BASELINE.json configs[1]: QAViT.py forward-only, synthetic 32x32x3, batch 512 (eval mode, bf16 autocast, one hipGraph)
if not sys.argv[1:] or "q32_fwd" in sys.argv[1:]:
    for variant in ("v1", "v2"):
        model = Q.QAViT(Q.qavit32_config(), variant); Q.fill_module(model); model = model.cuda().eval()
        g = torch.Generator().manual_seed(1234)
        x = torch.randn(512, 3, 32, 32, generator=g).cuda()
        def fwd():
            with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
                return model(x)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3): fwd()
        torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fwd()
        for _ in range(20): gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 50
        for _ in range(n): gr.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"q32_fwd {variant}: B=512 eval forward {dt*1e3:.3f} ms  {512/dt:.0f} img/s  finite {bool(torch.isfinite(out.float()).all())}", flush=True)
