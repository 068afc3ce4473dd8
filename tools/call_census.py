"""Every C-ABI call of ONE eager training step, in issue order, with its device time (bench.KernelTimer's event brackets) and the
shape arguments that identify it -- the map of the step's launch chain used to decide what to fuse next.
usage: call_census.py [batch] [config]      (on the GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import qavit_amd as Q  # noqa: E402
import importlib  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
config = sys.argv[2] if len(sys.argv) > 2 else "c100"
Q.lib.load()
cfg = Q.HQAViTConfig() if config == "c100" else Q.HQAViTTinyINConfig()
model = Q.HQAViT(cfg)
Q.fill_module(model)
model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, cfg.img_size, cfg.img_size, generator=g).cuda()
y = torch.randint(0, cfg.num_classes, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(batch_size=B, use_amp=True), total_steps=1000, warmup_steps=10, compute_dtype=torch.bfloat16)
for _ in range(3):
    tr.step(x, y)
torch.cuda.synchronize()
lib = importlib.import_module("qa-vit_amd.lib").load()


def describe(name, a):
    o = bench.KernelTimer._obj
    try:
        if name == "qavit_gemm_nt":
            g_ = o(a[0])
            return f"M={g_.M} N={g_.N} K={g_.K} a_mode={g_.a_mode} act={g_.act} R={'y' if g_.R else 'n'} Z={'y' if g_.Z else 'n'}"
        if name == "qavit_gemm_nt_grouped":
            return "n=%d: " % a[1] + " ".join(f"[{a[0][i].M}x{a[0][i].N}x{a[0][i].K}]" for i in range(min(a[1], 4)))
        if name in ("qavit_gemm_tn_grouped", "qavit_gemm_tn_grouped_ws"):
            return f"problems={a[1]}"
        if name == "qavit_layernorm_bwd":
            return f"rows={a[9]} C={a[10]} dres={'y' if a[15] else 'n'}"
        if name == "qavit_layernorm_fwd":
            return f"rows={a[6]} C={a[7]} add={'y' if a[10] else 'n'}"
        if name == "qavit_layernorm_bwd_sum":
            return f"n={a[1]} rows={a[11]} C={a[12]}"
        if name == "qavit_ln_param_reduce":
            return f"descs={a[1]}"
        if name in ("qavit_branch_fwd", "qavit_branch_bwd"):
            g_ = o(a[0])
            return f"kind={g_.kind} B={g_.B} T={g_.T}"
    except Exception as e:      # noqa: BLE001
        return f"({e})"
    return ""


class Census(bench.KernelTimer):
    def _wrap(self, name):
        orig = getattr(self.lib, name)
        self._orig[name] = orig

        def f(*a):
            torch.cuda._sleep(self.spin)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a)
            e1.record()
            self.rec.append((name, describe(name, a), torch.cuda.current_stream().cuda_stream, e0, e1))
            return r
        setattr(self.lib, name, f)


with Census(lib) as kt:
    tr.step(x, y)
    torch.cuda.synchronize()
    streams = {}
    tot = 0.0
    for i, (name, desc, st, e0, e1) in enumerate(kt.rec):
        us = max(e0.elapsed_time(e1) - kt.empty_ms, 0.0) * 1e3
        tot += us
        sid = streams.setdefault(st, len(streams))
        print(f"{i:4d} s{sid} {us:7.1f} us  {name.replace('qavit_', ''):28s} {desc}")
    print(f"{len(kt.rec)} calls, {tot / 1e3:.3f} ms of device time in brackets")
