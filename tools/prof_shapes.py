"""Per-shape device time of qavit_gemm_nt over one eager training step: HIP-event brackets behind a spin kernel (the
device stays busy while the host enqueues), empty-bracket cost subtracted -- the method of bench.py's KernelTimer."""
import sys, os, torch, importlib
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
L = importlib.import_module("qa-vit_amd.lib")
lib = L.load()
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(1024, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (1024,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(), total_steps=1000, warmup_steps=10)
for _ in range(2): tr.step(x, y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); torch.cuda._sleep(20000); e1.record(); torch.cuda.synchronize()
spin = max(int(30.0 / (e0.elapsed_time(e1) * 1e3 / 20000.0)), 1)
pairs = []
for _ in range(64):
    torch.cuda._sleep(spin); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); b.record(); pairs.append((a, b))
torch.cuda.synchronize()
empty = sorted(a.elapsed_time(b) for a, b in pairs)[32]
rec = []
orig = lib.qavit_gemm_nt
def wrapped(a, st):
    g_ = a._obj
    torch.cuda._sleep(spin)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); r = orig(a, st); e.record()
    full = bool(g_.Z or g_.act or g_.drop_p > 0 or g_.dp_p > 0 or g_.R)
    rec.append(((g_.dtype, g_.M, g_.N, g_.K, g_.a_mode, int(full)), s, e))
    return r
lib.qavit_gemm_nt = wrapped
tr.step(x, y)
torch.cuda.synchronize()
lib.qavit_gemm_nt = orig
agg = defaultdict(lambda: [0, 0.0])
for k, s, e in rec:
    d = agg[k]; d[0] += 1; d[1] += max(s.elapsed_time(e) - empty, 0.0)
tot = sum(v[1] for v in agg.values())
print(f"gemm_nt calls {len(rec)}, total {tot:.3f} ms, empty bracket {empty*1e3:.1f} us")
for k, (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    dt, M, N, Kd, am, full = k
    by = (M * Kd + N * Kd + M * N) * (2 if dt == 1 else 4)
    print(f"{'bf16' if dt==1 else 'f32 '} M={M:6d} N={N:4d} K={Kd:4d} am{am} epi{full} x{c:3d} {ms:7.3f} ms {1e3*ms/c:7.1f} us/call {2.0*M*N*Kd*c/ms/1e9:7.1f} TF/s {by*c/ms/1e9:6.2f} TB/s")
