"""Per-workgroup timeline of the one-launch weight-gradient kernel (diagnostic build: tools/build_stamps_lib.py gemm_tn_wide QAVIT_TN_STAMPS,
then QAVIT_LIB=qa-vit_amd/libqavit_stamps.so python3 tools/tn_stamps.py [B=1024]).  Records the dW problems one training step queues (as
tools/tn_census.py does), runs exactly that list through the one launch and reads back, per workgroup: start / end (s_memtime, 10 ns
ticks), segments run, and time + planned cost per tile class -- is the launch as long as its slowest range, and which class mis-prices?"""
import sys, os, importlib, statistics, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels"); F = importlib.import_module("qa-vit_amd.functional")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=True), total_steps=1000, warmup_steps=10, compute_dtype=torch.bfloat16)
tr.step(x, y)
rec = []
orig = K.gemm_tn
def spy(A, Bm, Cgrad, M, N, Kd, lda, ldb, ldc, colsum=None, ln=None, **kw):
    if K.DeferredTN.enabled and A.dtype == torch.bfloat16:
        rec.append((M, N, Kd, ln is not None, colsum is not None or "colsum_ptr" in kw))
    return orig(A, Bm, Cgrad, M, N, Kd, lda, ldb, ldc, colsum, ln, **kw)
K.gemm_tn = spy; F.K.gemm_tn = spy
tr.step(x, y); torch.cuda.synchronize()
K.gemm_tn = orig; F.K.gemm_tn = orig
del tr, model
dev, dt = "cuda", torch.bfloat16
probs = []
for (M, N, Kd, ln, cs) in rec:
    A = torch.randn(M, N, device=dev).to(dt); Bm = torch.randn(M, Kd, device=dev).to(dt)
    Cg = torch.zeros(N, Kd, device=dev); csum = torch.zeros(N, device=dev) if cs else None
    lnarg = (torch.ones(Kd, device=dev), torch.zeros(Kd, device=dev), torch.zeros(M, device=dev), torch.ones(M, device=dev)) if ln else None
    probs.append((A, Bm, Cg, csum, lnarg, M, N, Kd))
tot_mb = sum(M * (N + Kd) * 2 for (M, N, Kd, _, _) in rec) / 1e6
print(f"{len(rec)} bf16 problems, {tot_mb:.1f} MB algorithmic operand bytes")
def fn():
    K.DeferredTN.enabled = True; K.DeferredTN.home_stream = None
    for (A, Bm, Cg, csum, lnarg, M, N, Kd) in probs:
        K.gemm_tn(A, Bm, Cg, M, N, Kd, N, Kd, Kd, csum, ln=lnarg)
    K.DeferredTN.flush(); K.DeferredTN.enabled = False
for _ in range(2):
    fn(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); fn(); e1.record(); torch.cuda.synchronize()
print(f"flush (table writes + the launch), events: {e0.elapsed_time(e1) * 1e3:.1f} us")
lib = Q.lib.load()
nwg = 256
buf = (C.c_ulonglong * (nwg * 48))()
lib.qavit_tn_stamps.restype = C.c_int
assert lib.qavit_tn_stamps(buf, nwg) == 0
rows = [[buf[w * 48 + k] for k in range(48)] for w in range(nwg)]
rows = [r for r in rows if r[0] and r[1]]
busy = sorted((r[1] - r[0]) for r in rows)
print(f"busy ticks per workgroup (each XCD has its own counter: only differences inside a workgroup mean anything): min {busy[0]} median {busy[len(busy) // 2]} "
      f"max {busy[-1]}  max/mean {busy[-1] / statistics.mean(busy):.3f}")
t0 = min(r[0] for r in rows)
starts = sorted((r[0] - t0) / 100 for r in rows); ends = sorted((r[1] - t0) / 100 for r in rows)
q = lambda v, p: v[min(len(v) - 1, int(p * len(v)))]
print(f"{len(rows)} workgroups; start us: median {q(starts, .5):.1f} max {starts[-1]:.1f};  end us: min {ends[0]:.1f} 10% {q(ends, .1):.1f} median {q(ends, .5):.1f} "
      f"90% {q(ends, .9):.1f} max {ends[-1]:.1f};  mean busy {statistics.mean((r[1] - r[0]) / 100 for r in rows):.1f}")
print(f"segments per workgroup: median {statistics.median(r[2] for r in rows)} max {max(r[2] for r in rows)}")
ins, jns = (1, 2, 4, 6, 8), (1, 2, 3, 4)
print("class (N tile x K tile)   time share   ticks per cost unit (10 ns; equal = the cost model prices the class right)")
tt = sum(sum(r[4:24]) for r in rows)
for ci in range(20):
    t = sum(r[4 + ci] for r in rows); c = sum(r[24 + ci] for r in rows)
    if c:
        print(f"  {32 * ins[ci // 4]:4d} x {64 * jns[ci % 4]:4d}   {100 * t / tt:6.1f} %   {t / c:8.4f}")
late = sorted(rows, key=lambda r: -r[1])[:6]
for r in late:
    cls = [(32 * ins[ci // 4], 64 * jns[ci % 4], round(r[4 + ci] / 100, 1)) for ci in range(20) if r[4 + ci]]
    print("late workgroup: end", round((r[1] - t0) / 100, 1), "us, classes (N, K, us):", cls)
allrows = [[buf[w * 48 + k] for k in range(48)] for w in range(nwg)]
print("mean busy ticks by XCD (blockIdx % 8):", [round(statistics.mean(r[1] - r[0] for i, r in enumerate(allrows) if i % 8 == x and r[1])) for x in range(8)])
print("mean busy ticks by position (32 workgroups each):", [round(statistics.mean(r[1] - r[0] for r in allrows[o:o + 32] if r[1])) for o in range(0, nwg, 32)])
print("mean busy ticks by segments run:", {k: round(statistics.mean(r[1] - r[0] for r in allrows if r[1] and r[2] == k)) for k in sorted({r[2] for r in allrows if r[1]})})
top = sorted(range(nwg), key=lambda i: -(allrows[i][1] - allrows[i][0]))[:10]
for i in top:
    r = allrows[i]
    print("slow workgroup", i, "busy", r[1] - r[0], "segments", r[2], [(32 * ins[ci // 4], 64 * jns[ci % 4], r[4 + ci]) for ci in range(20) if r[4 + ci]])
