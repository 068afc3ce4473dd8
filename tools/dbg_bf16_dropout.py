"""Debug aid: where do the bf16 and fp32 paths of one HQA-ViT train-mode forward part ways?  Same module (same dropout sites), same step."""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
Q.lib.load()
for dropout, dp in ((0.0, 0.0), (0.1, 0.0), (0.0, 0.1), (0.1, 0.1)):
    cfg = Q.HQAViTConfig(dropout=dropout, drop_path=dp)
    m = Q.HQAViT(cfg); Q.fill_module(m)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m = m.cuda().train()
    if dropout == 0:
        for mm in m.modules():
            if isinstance(mm, torch.nn.Dropout): mm.p = 0.0
    g = torch.Generator().manual_seed(21)
    x = torch.randn(24, 3, 32, 32, generator=g).cuda()
    res = {}
    for dtype in (torch.float32, torch.bfloat16):
        m.load_state_dict(sd)
        taps, hooks = {}, []
        for n, mod in m.named_modules():
            if n and n.count(".") <= 3:
                def hk(mo, i, o, n=n):
                    t = o[0] if isinstance(o, tuple) else o
                    if torch.is_tensor(t): taps[n] = t.detach().float().cpu()
                hooks.append(mod.register_forward_hook(hk))
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == torch.bfloat16):
            out = m(x)
        for h in hooks: h.remove()
        res[dtype] = (out.detach().float().cpu(), taps)
    (o32, t32), (o16, t16) = res[torch.float32], res[torch.bfloat16]
    print("dropout", dropout, "dp", dp, "logits maxrel", float((o16 - o32).abs().max() / o32.abs().max()), "rms-rel", float((o16 - o32).norm() / o32.norm()))
    k = 0
    for n in t32:
        if n in t16 and t32[n].shape == t16[n].shape:
            a, b = t16[n], t32[n]
            mr, rr, mm_ = float((a - b).abs().max() / b.abs().max().clamp_min(1e-20)), float((a - b).norm() / b.norm().clamp_min(1e-20)), float(((a == 0) != (b == 0)).float().mean())
            if (rr > 0.02 or mm_ > 0.001) and k < 14:
                print("    %-50s max-rel %.3f rms-rel %.4f mask-mismatch %.4f" % (n, mr, rr, mm_)); k += 1
