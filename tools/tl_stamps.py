"""In-kernel timeline of the fused TokenLearner backward (diagnostic build: tools/build_stamps_lib.py tokens_tl QAVIT_TL_STAMPS, then
QAVIT_LIB=qa-vit_amd/libqavit_stamps.so python3 tools/tl_stamps.py [B=1024]).  s_memtime runs at 100 MHz: 10 ns per tick."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import qavit_amd as Q  # noqa: E402
from importlib import import_module  # noqa: E402

F = import_module("qa-vit_amd.functional")
M_ = import_module("qa-vit_amd.modules")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
lib = Q.lib.load()
tl = M_.TokenLearner(192, 16).cuda()
x = torch.randn(B, 64, 192, device="cuda").to(torch.bfloat16).requires_grad_(True)
go = torch.randn(B, 16, 192, device="cuda").to(torch.bfloat16)
for _ in range(3):
    x.grad = None
    tl(x).backward(go)
torch.cuda.synchronize()
nwg = 256
buf = (C.c_ulonglong * (nwg * 40))()
lib.qavit_tl_stamps.restype = C.c_int
assert lib.qavit_tl_stamps(buf, nwg) == 0
names = {0: "start", 1: "weight staged + sync", 36: "images done", 37: "end (after flush)"}
for k in range(4):
    names.update({2 + 8 * k: f"img{k}: rows / dxc / P staged (stores issued)", 3 + 8 * k: f"img{k}: sync A passed", 4 + 8 * k: f"img{k}: dP + partial dots",
                  5 + 8 * k: f"img{k}: sync B passed, dz written", 6 + 8 * k: f"img{k}: sync C passed", 7 + 8 * k: f"img{k}: dxn MFMAs",
                  8 + 8 * k: f"img{k}: LayerNorm backward + mix + dx stores issued", 9 + 8 * k: f"img{k}: dW MFMAs"})
import statistics  # noqa: E402
rows = [[buf[w * 40 + k] for k in range(40)] for w in range(nwg)]
rows = [r for r in rows if r[0] and r[37]]
prev = None
for k in sorted(names):
    d = [r[k] - r[0] for r in rows if r[k]]
    if not d:
        continue
    med = statistics.median(d)
    print(f"{k:3d} {names[k]:60s} {med / 100:8.2f} us" + (f"   (+{(med - prev) / 100:6.2f})" if prev is not None else ""))
    prev = med
