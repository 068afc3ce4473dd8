"""In-kernel phase timeline of the fused branch forward (diagnostic build only):
    QAVIT_EXTRA_HIPCC_FLAGS=-DQAVIT_BRANCH_STAMPS python qa-vit_amd/build.py --force && python tools/branch_stamps.py [kind]
then rebuild without the flag.  The stamped kernel writes s_memtime ticks (shader cycles) per workgroup into a device buffer of its own
(read back through the diagnostic build's qavit_branch_stamps).  usage: branch_stamps.py [kind] [save: 0 | 1]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
import qavit_amd as Q
F = import_module("qa-vit_amd.functional"); K = import_module("qa-vit_amd.kernels")
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 0
save = len(sys.argv) > 2 and sys.argv[2] == "1"
B, T, C, S, KC = 1024, 16, 192, 16, 32
g = torch.Generator().manual_seed(0)
def rnd(*s, sc=1.0): return (torch.randn(*s, generator=g) * sc).cuda()
x = rnd(B, T, C).to(torch.bfloat16)
nq = C if kind == 2 else 3 * C
wqkv, bqkv, wproj, bproj = rnd(nq, C, sc=0.05), rnd(nq, sc=0.1), rnd(C, C, sc=0.05), rnd(C, sc=0.1)
Ek, Ev = rnd(128, KC, sc=0.3), rnd(128, KC, sc=0.3)
bk, bv = rnd(S, C, sc=0.5), rnd(S, C, sc=0.5)
t = [y * 4 + xx for d in (1, 2) for y in range(0, 4, d) for xx in range(0, 4, d)]
idx = torch.tensor(t, dtype=torch.int32, device="cuda")
for it in range(3):
    res = F.branch_forward(kind, x, wqkv, bqkv, wproj, bproj, None if kind == 2 else Ek, None if kind == 2 else Ev, bk, bv,
                           idx if kind == 1 else None, 2 if kind == 1 else 0, (16, 10, 0)[kind], (0.1, 1), (0.1, 2), want_o=True, save=save)
torch.cuda.synchronize()
import ctypes
lib = Q.lib.load()
host = (ctypes.c_uint64 * (256 * 16))()
assert lib.qavit_branch_stamps(host, 256) == 0
st = torch.tensor(list(host), dtype=torch.int64).reshape(256, 16)[:, :12].double()
print(f"kind {kind} save {save}")
d = st - st[:, :1]
names = ["start (4 chunks issued)", "constants staged", "x tiles staged", "barrier (all landed)", "q phase done (6 chunks)", "k phase + Kf done",
         "S + softmax done (4 images)", "v phase + Vf done", "O^T done, O quads in LDS", "proj GEMM done (6 chunks)", "output rows stored", "end"]
prev = 0.0
for k in range(12):
    m = d[:, k].median().item()
    print(f"  {names[k]:32s} median {m:9.0f}  (+{m - prev:7.0f})   max {d[:, k].max().item():9.0f}")
    prev = m
