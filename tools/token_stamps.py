"""In-kernel timeline of the TokenUpMix backward (diagnostic build: tools/build_stamps_lib.py tokens_bf16 QAVIT_TOKEN_STAMPS, then
QAVIT_LIB=qa-vit_amd/libqavit_stamps.so python3 tools/token_stamps.py [B=1024] [N=64] [M=16]).  s_memtime runs at 100 MHz: 10 ns per tick."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import qavit_amd as Q  # noqa: E402
from importlib import import_module  # noqa: E402

F = import_module("qa-vit_amd.functional")
K = import_module("qa-vit_amd.kernels")
L = import_module("qa-vit_amd.lib")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
M = int(sys.argv[3]) if len(sys.argv) > 3 else 16
lib = Q.lib.load()
dev, dt, Cc = "cuda", torch.bfloat16, 192
g = torch.Generator().manual_seed(5)
xc = torch.randn(B, M, Cc, generator=g).to(dev).to(dt).requires_grad_(True)
W = (torch.randn(N, M, generator=g) * 0.2).to(dev).requires_grad_(True)
bias, gam, bet = (torch.zeros(N, device=dev, requires_grad=True), torch.ones(Cc, device=dev, requires_grad=True), torch.zeros(Cc, device=dev, requires_grad=True))
for p in (W, bias, gam, bet):
    p.grad = torch.zeros_like(p)
gy = torch.randn(B, N, Cc, generator=g).to(dev).to(dt)
for _ in range(3):
    xc.grad = None
    K.DeferredLN.enabled = True
    F.UpMixFn.apply(xc, W, bias, gam, bet, 1e-5).backward(gy)
    K.DeferredLN.flush()
    K.DeferredLN.enabled = False
torch.cuda.synchronize()
nwg = 512
buf = (C.c_ulonglong * (nwg * 32))()
lib.qavit_token_stamps.restype = C.c_int
assert lib.qavit_token_stamps(buf, nwg) == 0
names = {0: "start", 1: "W staged, first loads requested", 30: "images done", 31: "end (after flush)"}
for k in range(4):
    names.update({2 + 6 * k: f"img{k}: top sync passed", 3 + 6 * k: f"img{k}: xc / dy committed, next requested", 4 + 6 * k: f"img{k}: up tile recomputed",
                  5 + 6 * k: f"img{k}: LayerNorm backward done, du in LDS", 6 + 6 * k: f"img{k}: second sync passed", 7 + 6 * k: f"img{k}: dxc + dW products done"})
for wg in (0, 1, 300):
    row = buf[wg * 32:(wg + 1) * 32]
    t0 = row[0]
    print(f"-- workgroup {wg}")
    for k in sorted(names):
        if row[k] >= t0 and row[k] - t0 < 10**7:
            print(f"   {(row[k] - t0) * 0.01:8.2f} us  {names[k]}")
