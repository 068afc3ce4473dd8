import sys, os, torch, importlib, ctypes as C, statistics
sys.path.insert(0, "/root/repo")
import qavit_amd as Q
K = importlib.import_module("qa-vit_amd.kernels")
dev = "cuda"; dt = torch.bfloat16
lib = Q.lib.load()
lib.qavit_tn_stamps.restype = C.c_int
for (M, N, Kd, cnt, ln) in [(65536, 192, 192, 16, False), (65536, 256, 256, 12, False), (65536, 64, 64, 48, False), (65536, 128, 128, 24, False), (16384, 48, 192, 100, True), (65536, 256, 1024, 5, False)]:
    probs = []
    for i in range(cnt):
        A = torch.randn(M, N, device=dev).to(dt); B = torch.randn(M, Kd, device=dev).to(dt)
        Cg = torch.zeros(N, Kd, device=dev); cs = torch.zeros(N, device=dev)
        lnarg = (torch.ones(Kd, device=dev), torch.zeros(Kd, device=dev), torch.zeros(M, device=dev), torch.ones(M, device=dev)) if ln else None
        probs.append((A, B, Cg, cs, lnarg))
    def fn():
        K.DeferredTN.enabled = True; K.DeferredTN.home_stream = None
        for (A, B, Cg, cs, lnarg) in probs:
            K.gemm_tn(A, B, Cg, M, N, Kd, N, Kd, Kd, cs, ln=lnarg)
        K.DeferredTN.flush(); K.DeferredTN.enabled = False
    for _ in range(3):
        fn(); torch.cuda.synchronize()
    buf = (C.c_ulonglong * (256 * 48))()
    assert lib.qavit_tn_stamps(buf, 256) == 0
    rows = [[buf[w * 48 + k] for k in range(48)] for w in range(256)]
    rows = [r for r in rows if r[1]]
    busy = statistics.mean(r[1] - r[0] for r in rows)
    ph = [statistics.mean(r[k] for r in rows) for k in (44, 45, 46, 47, 3)]
    print(f"M={M} N={N} K={Kd} ln={int(ln)}: busy {busy:.0f} ticks; load wait {100*ph[0]/busy:.1f}% barrierA {100*ph[1]/busy:.1f}% staging {100*ph[2]/busy:.1f}% barrierB {100*ph[3]/busy:.1f}% mfma {100*ph[4]/busy:.1f}%  rest {100*(busy-sum(ph))/busy:.1f}%", flush=True)
    del probs
