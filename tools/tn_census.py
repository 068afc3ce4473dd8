"""Which weight-gradient GEMMs does one training step queue, and how long do exactly those take?

Records the (M, N, K, LayerNorm-on-load) of every problem the autograd layer hands to qavit_gemm_tn / _grouped in one eager
step at B images, prints the histogram, then times the same list (fresh random operands, flushed in the step's group sizes)
under a hipGraph.  The number to compare with bench.py's `gemm_tn_grouped` + `gemm_tn` kernel time."""
import sys, os, collections, importlib
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels"); L = importlib.import_module("qa-vit_amd.lib")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=True), total_steps=1000, warmup_steps=10, compute_dtype=torch.bfloat16)
tr.step(x, y)
rec = []
orig = K.gemm_tn
def spy(A, Bm, Cgrad, M, N, Kd, lda, ldb, ldc, colsum=None, ln=None, **kw):
    deferred = K.DeferredTN.enabled
    rec.append((M, N, Kd, ln is not None, colsum is not None or "colsum_ptr" in kw, deferred, str(A.dtype)))
    return orig(A, Bm, Cgrad, M, N, Kd, lda, ldb, ldc, colsum, ln, **kw)
K.gemm_tn = spy
F = importlib.import_module("qa-vit_amd.functional"); F.K.gemm_tn = spy
tr.step(x, y); torch.cuda.synchronize()
K.gemm_tn = orig; F.K.gemm_tn = orig
hist = collections.Counter(rec)
tot_b = 0
for (M, N, Kd, ln, cs, d, dt), c in sorted(hist.items(), key=lambda kv: -kv[1] * kv[0][0] * (kv[0][1] + kv[0][2])):
    mb = M * (N + Kd) * 2 / 1e6
    tot_b += mb * c
    print(f"x{c:3d}  M={M:6d} N={N:4d} K={Kd:4d} ln={int(ln)} colsum={int(cs)} deferred={int(d)} {dt}  {mb:7.2f} MB each")
print(f"{len(rec)} problems, {tot_b:.1f} MB algorithmic operand bytes")

dev = "cuda"; dt = torch.bfloat16
probs = []
for (M, N, Kd, ln, cs, d, _) in rec:
    if not d:
        continue
    A = torch.randn(M, N, device=dev).to(dt); Bm = torch.randn(M, Kd, device=dev).to(dt)
    Cg = torch.zeros(N, Kd, device=dev); csum = torch.zeros(N, device=dev) if cs else None
    lnarg = None
    if ln:
        mean, rstd = torch.zeros(M, device=dev), torch.ones(M, device=dev)
        lnarg = (torch.ones(Kd, device=dev), torch.zeros(Kd, device=dev), mean, rstd)
    probs.append((A, Bm, Cg, csum, lnarg, M, N, Kd))
def fn():
    K.DeferredTN.enabled = True; K.DeferredTN.home_stream = None
    for (A, Bm, Cg, csum, lnarg, M, N, Kd) in probs:
        K.gemm_tn(A, Bm, Cg, M, N, Kd, N, Kd, Kd, csum, ln=lnarg)
    K.DeferredTN.flush(); K.DeferredTN.enabled = False
fn(); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph(); s = torch.cuda.Stream()
with torch.cuda.stream(s):
    fn()
    with torch.cuda.graph(gr, stream=s):
        fn()
for _ in range(3):
    gr.replay()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    gr.replay()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
db = sum(M * (N + Kd) * 2 for (*_, M, N, Kd) in probs) / 1e9
fl = sum(2.0 * M * N * Kd for (*_, M, N, Kd) in probs) / 1e12
print(f"deferred problems of one step: {len(probs)}  {ms:.3f} ms  {db / ms * 1e3:.0f} GB/s algorithmic  {fl / ms * 1e3:.1f} TFLOP/s")
