#!/usr/bin/env python3
"""Benchmark of the hot path BASELINE.json names: HQA-ViT CIFAR-100 training step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one full training step of HQAViT(HQAViTConfig()) on a synthetic 32x32x3 batch of 1024 images per
GPU (BASELINE.json configs[2]/[3]): weight re-pack, forward, loss, backward, gradient all-reduce (N > 1),
clipping, fused AdamW and the EMA update -- nothing is skipped inside the timed region.  bf16 activations, fp32 accumulate
and master weights; reference-default dropout (the proj / MLP dropouts AND the dropout_p the reference hands to SDPA) and drop-path.  Inputs are resident in HBM before timing starts.

Rank 0 prints ONE JSON line with the contract fields plus
  roofline     : the dominant kernel family of the step, timed per launch with HIP events on the launch stream
                 during one instrumented (eager) step of this same workload;
  cpu_baseline : the CPU oracle (oracle/qavit_oracle.py, a restatement = "port") timed on the host cores on a
                 bounded sample (B=32 train steps), N=1 only.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFLOP_PER_IMG_TRAIN = 1293.0       # fwd+bwd work AS THE REFERENCE COMPUTES IT, HQA-ViT CIFAR-100
# Work the path actually NEEDS (utilisation must not be inflated by work that is skipped).  Per block and image the
# reference computes and discards: 2/3 of MSDA's qkv(x) (2.36 MFLOP), 118 all-zero rows of the 128-row Linformer product (2.90),
# cross-attention's k/v projections of the batch-invariant bank once per image (2.36) = 7.62 MFLOP x 8 blocks = 61 forward,
# x3 for forward + backward = 183; and the train-mode bank write applies Linear(192,192) to every token where only the 16 mean rows
# need it (35.4 -> 7.1 over the 24 writes: 28.3).  1293 - 183 - 28 = 1082.
NEEDED_MFLOP_PER_IMG_TRAIN = 1082.0
# needed forward work of one fused attention branch per image (SURVEY.md 8d per-block table): qkv + Linformer + SDPA + proj
BRANCH_MFLOP_PER_IMG = {0: 3.54 + 0.39 + 0.59 + 1.18, 1: 1.18 + 2.21 + 0.25 + 0.59 + 1.18, 2: 1.18 + 0.20 + 1.18}
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=30, help="untimed steps; a fresh box needs ~0.5 s of replays before its clocks and caches settle")
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--config", default="c100", choices=["c100", "tin", "q32"], help="c100 = the BASELINE metric's model (default); tin = HQAViT_IN_Tiny defaults (64x64, 200 classes); "
                    "q32 = BASELINE config 2: QAViT.py at 32 px (N = 64 tokens, no TokenLearner), forward only (use with --eval, --batch 512)")
    ap.add_argument("--eval", action="store_true", help="forward-only in eval mode (q32: BASELINE config 2)")
    ap.add_argument("--variant", default="v1", choices=["v1", "v2"], help="q32: QAViT.py (v1) or QAViTv2.py (v2) block variant")
    ap.add_argument("--mix", action="store_true", help="include device-side CutMix/MixUp + mixed loss in the step (off: the BASELINE metric)")
    ap.add_argument("--fwd-bwd-only", action="store_true", help="time forward+backward(+all-reduce) without the optimiser")
    ap.add_argument("--ddp-tags", default=None, help="data-parallel sync points inside backward: 'none' (one bucketed all-reduce after backward), "
                    "'fuse3' (one overlapped bucket prefix), 'all' (seven), or a comma list of tags; default: parallel.DataParallel's own (by world size)")
    return ap.parse_args()


def spawn_ranks(args):
    """``python bench.py --gpus N`` started directly (no WORLD_SIZE in the environment): THIS process has made no GPU call yet, so it may
    start the N ranks as children -- ``python -m torch.distributed.run --nproc-per-node N bench.py <same flags>`` -- relay what they print
    (rank 0's JSON line on stdout) and exit with the launcher's code (non-zero if any rank failed).  Never an exec: a process that has
    touched the GPU must not be replaced, and this one stays the parent."""
    import socket
    import subprocess
    n_dev = torch.cuda.device_count()                       # counting devices does not initialise the GPU on this image
    if n_dev < args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible on this node")
    with socket.socket() as sk:                             # a free rendezvous port on the loopback address
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # RCCL over dmabuf IPC (the host driver supports nothing else)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: starting the ranks myself: {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    rc = subprocess.run(cmd, env=env).returncode
    raise SystemExit(rc)


# ---------------------------------------------------------------------------------------------------
# per-launch kernel timing with HIP events (roofline)
# ---------------------------------------------------------------------------------------------------
class KernelTimer:
    """Times every C-ABI entry point of libqavit_hip.so during one eager step of the benchmarked workload.

    Each call is bracketed by a torch.cuda.Event pair recorded on the launch stream (the kernels are enqueued on
    torch's current stream).  An eager step is launch-bound on the host, so a short spin kernel is queued in front of
    each bracket: the GPU is still busy with it while the host enqueues start-event, kernel(s), stop-event, which then
    run back to back on the device -- the bracket measures device time, not host launch latency.
    Algorithmic flops / bytes per call come from the call's own arguments (formulas below)."""

    ESZ = {0: 4, 1: 2}

    def __init__(self, lib):
        self.lib, self.rec, self._orig = lib, [], {}
        # calibrate the spin kernel (wall-clock ticks on ROCm) to ~30 us
        torch.cuda._sleep(1000); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); torch.cuda._sleep(20000); e1.record(); torch.cuda.synchronize()
        us_per_tick = max(e0.elapsed_time(e1) * 1e3 / 20000.0, 1e-6)
        self.spin = max(int(30.0 / us_per_tick), 1)
        # cost of an empty bracket (start-event, stop-event back to back behind the spin): subtracted from every call
        pairs = []
        for _ in range(64):
            torch.cuda._sleep(self.spin)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); b.record()
            pairs.append((a, b))
        torch.cuda.synchronize()
        ts = sorted(a.elapsed_time(b) for a, b in pairs)
        self.empty_ms = ts[len(ts) // 2]

    @staticmethod
    def _obj(x):
        return getattr(x, "_obj", x)

    def _work(self, name, a):
        """-> (flops, bytes) of one call; bytes = operands read once + results written once."""
        esz = self.ESZ
        if name == "qavit_gemm_nt":
            g = self._obj(a[0]); e = esz[g.dtype]
            by = (g.M * g.K + g.N * g.K + g.M * g.N) * e
            if g.a_mode == 2:
                by += g.M * g.K * e * ((1 if g.a_Z else 0) + (1 if g.a_out else 0))
            if g.Z: by += g.M * g.N * e
            if g.R: by += g.M * g.N * e
            return 2.0 * g.M * g.N * g.K, float(by)
        if name in ("qavit_gemm_tn", "qavit_gemm_tn_grouped"):
            probs = [self._obj(a[0])] if name == "qavit_gemm_tn" else [a[0][i] for i in range(a[1])]
            fl = sum(2.0 * g.M * g.N * g.K for g in probs)
            by = sum((g.M * (g.N + g.K)) * esz[g.dtype] + g.N * g.K * 4 for g in probs)
            return fl, float(by)
        if name in ("qavit_attn_fwd", "qavit_attn_bwd"):
            g = self._obj(a[0]); e = esz[g.dtype]
            nk = (g.KC if g.mode == 0 else g.L) + g.S
            fl = 4.0 * g.G * g.H * g.Nq * nk * g.D + (4.0 * g.G * g.H * g.KC * g.L * g.D if g.mode == 0 else 0.0)
            by = (g.G * g.Nq + 2 * g.G * g.L) * g.H * g.D * e + g.G * g.Nq * g.H * g.D * e
            return (fl, float(by)) if name == "qavit_attn_fwd" else (2.5 * fl, 2.5 * by)
        if name == "qavit_branch_fwd":
            g = self._obj(a[0])
            by = g.B * g.T * g.C * 2 * (3 if g.o_save else 2) + (4 if g.kind != 2 else 2) * g.C * g.C * 2
            return BRANCH_MFLOP_PER_IMG[g.kind] * 1e6 * g.B, float(by)
        if name in ("qavit_compress_fuse_fwd", "qavit_compress_fuse_bwd"):
            # four Linear(192 -> 48) per token row (+ their input-gradient GEMMs backward); fwd: 4 x in, cat + y out; bwd: dy, cat, 4 x in, dcat + 4 dx out
            g = self._obj(a[0])
            rows = g.B * g.T
            fl = 2.0 * rows * g.NB * g.CB * g.C
            if name == "qavit_compress_fuse_fwd":
                return fl, float(rows * (g.NB * g.C + 2 * g.NB * g.CB) * 2)
            return fl, float(rows * (2 * g.NB * g.C + 3 * g.NB * g.CB) * 2)
        if name in ("qavit_cga_fwd", "qavit_cga_bwd"):
            # per image: 6 groups x (q/k/v projections 2*16*32*48 + 4 heads x (QK^T + PV over 32 keys, D = 4)) + proj 2*16*96*192; x in, out (+ o) out;
            # backward ~2.5x the forward's flops, dout + x in, dz + dqkv + dx out
            g = self._obj(a[0])
            fwd = g.B * (g.G * (2.0 * g.T * 32 * 48 + g.H * 4.0 * g.T * 32 * g.D) + 2.0 * g.T * 96 * g.C)
            if name == "qavit_cga_fwd":
                return fwd, float(g.B * g.T * (2 * g.C + 96) * 2)
            return 2.5 * fwd, float(g.B * g.T * (4 * g.C + 6 * 48) * 2)
        if name == "qavit_branch_bwd":
            # proj input gradient (2 T C C) + attention-core backward (~2.5x the forward core, as for qavit_attn_bwd); operands read / written once:
            # dout, q, o in; dz, dq out; k, v in and dk, dv out for SWA / MSDA
            g = self._obj(a[0])
            nk = (g.KC if g.kind != 2 else 0) + g.S
            core = 4.0 * g.H * g.T * nk * g.D + (4.0 * g.H * g.KC * g.L * g.D if g.kind != 2 else 0.0)
            fl = g.B * (2.0 * g.T * g.C * g.C + 2.5 * core)
            by = g.B * g.T * g.C * 2 * 5 + (4 * g.B * g.kv_rows * g.C * 2 if g.kind != 2 else 0) + g.C * g.C * 2
            return fl, float(by)
        if name in ("qavit_layernorm_fwd", "qavit_layernorm_bwd", "qavit_row_stats"):
            e = esz[a[0]]
            if name == "qavit_layernorm_fwd": rows, Cc, k = a[6], a[7], 2
            elif name == "qavit_layernorm_bwd": rows, Cc, k = a[9], a[10], 3
            else: rows, Cc, k = a[3], a[4], 1
            return 8.0 * rows * Cc, float(k * rows * Cc * e)
        return 0.0, 0.0

    def _wrap(self, name):
        orig = getattr(self.lib, name)
        self._orig[name] = orig

        def f(*a):
            torch.cuda._sleep(self.spin)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a)
            e1.record()
            fl, by = self._work(name, a)
            self.rec.append((name, fl, by, e0, e1))
            return r
        setattr(self.lib, name, f)

    def __enter__(self):
        import importlib
        L = importlib.import_module("qa-vit_amd.lib")
        for name in L.EXPORTS:
            if name in ("qavit_version", "qavit_last_error", "qavit_attn_ws_floats", "qavit_bank_ws_floats", "qavit_branch_supported",
                        "qavit_layernorm_bwd_parts", "qavit_branch_bwd_parts", "qavit_cga_supported", "qavit_cga_bwd_parts", "qavit_ccf_bwd_parts", "qavit_compress_fuse_supported", "qavit_compress_fuse_bwd_parts"):      # host-side queries: nothing is launched
                continue
            self._wrap(name)
        return self

    def __exit__(self, *exc):
        for n, o in self._orig.items():
            setattr(self.lib, n, o)

    def summary(self):
        torch.cuda.synchronize()
        fam = {}
        for name, fl, by, e0, e1 in self.rec:
            d = fam.setdefault(name.replace("qavit_", ""), dict(launches=0, ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["ms"] += max(e0.elapsed_time(e1) - self.empty_ms, 0.0)
            d["flops"] += fl
            d["bytes"] += by
        return fam


def cpu_baseline(batch=32, budget_s=20.0, tin=False):
    """CPU oracle (a port of the reference's PyTorch-CPU path) timed on this host: one HQA-ViT train step of the benched configuration."""
    import qavit_amd as Q
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import qavit_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    # a 1-GPU box exposes a 16-core CPU share whatever the affinity mask says; oversubscribing it stalls for minutes
    cores = max(1, min(cores, int(os.environ.get("QAVIT_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    print(f"[bench] cpu baseline on {cores} threads ...", file=sys.stderr, flush=True)
    cfg = Q.HQAViTTinyINConfig() if tin else Q.HQAViTConfig()
    if tin:
        batch = 8
    model = Q.HQAViT(cfg)
    Q.fill_module(model)
    P = {k: v.clone() for k, v in model.state_dict().items()}
    for k in list(P):
        if k.endswith(("global_bank.global_k", "global_bank.global_v", "global_bank.update_count")):
            P[k] = P["global_bank." + k.rsplit(".", 1)[-1]]
    for n, _ in model.named_parameters():
        P[n].requires_grad_(True)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, 3, cfg.img_size, cfg.img_size, generator=g)
    y = torch.randint(0, cfg.num_classes, (batch,), generator=g)
    times = []
    t_start = time.time()
    it = 0
    while True:
        t0 = time.time()
        loss = O.loss_fn(O.hqavit_forward(P, x, cfg, train=True), y)
        loss.backward()
        for n, _ in model.named_parameters():
            P[n].grad = None
        dt = time.time() - t0
        if it >= 2:
            times.append(dt)
        it += 1
        if (time.time() - t_start > budget_s and len(times) >= 3) or len(times) >= 12:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(batch / med, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"HQA-ViT {'Tiny-ImageNet' if tin else 'CIFAR-100'} train fwd+bwd, B={batch}, fp32, {len(times)} timed steps after 2 warm-up, median {med * 1e3:.0f} ms/step"}


QA32_MFLOP_PER_IMG_EVAL = 709.9    # QA-ViT (img 32, patch 4, window 4, dilations (1,2), linformer_k 32) eval forward


def cpu_baseline_q32(variant, budget_s=20.0):
    """CPU oracle timed on this host: QA-ViT@32 eval forward on a bounded sample (B = 64)."""
    import qavit_amd as Q
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import qavit_oracle as O
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    cores = max(1, min(cores, int(os.environ.get("QAVIT_CPU_THREADS", "16"))))
    torch.set_num_threads(cores)
    cfg = Q.qavit32_config()
    model = Q.QAViT(cfg, variant)
    Q.fill_module(model)
    P = {k: v.clone() for k, v in model.state_dict().items()}
    for k in list(P):
        if k.endswith(("global_bank.global_k", "global_bank.global_v", "global_bank.update_count")):
            P[k] = P["global_bank." + k.rsplit(".", 1)[-1]]
    g = torch.Generator().manual_seed(1234)
    batch = 64
    x = torch.randn(batch, 3, 32, 32, generator=g)
    times, t_start, it = [], time.time(), 0
    with torch.no_grad():
        while True:
            t0 = time.time()
            O.qavit_forward(P, x, cfg, train=False, variant=variant)
            dt = time.time() - t0
            if it >= 2:
                times.append(dt)
            it += 1
            if (time.time() - t_start > budget_s and len(times) >= 3) or len(times) >= 12:
                break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(batch / med, 2), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": f"QA-ViT@32 ({variant}) eval forward, B={batch}, fp32, {len(times)} timed passes after 2 warm-up, median {med * 1e3:.0f} ms"}


def bench_q32_eval(args):
    """BASELINE.json configs[1]: QAViT.py forward-only on a synthetic 32x32x3 batch of 512, one GPU, eval mode, one hipGraph per pass."""
    import qavit_amd as Q
    Q.lib.load()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    B = args.batch
    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = Q.qavit32_config()
    model = Q.QAViT(cfg, args.variant)
    Q.fill_module(model)
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, 32, 32, generator=g).to(dev)

    def fwd():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16, enabled=cdt == torch.bfloat16):
            return model(x)
    mode = "eager"
    run = fwd
    out = None
    if not args.no_graph:
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(3):
                fwd()
        torch.cuda.current_stream().wait_stream(s_)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            out = fwd()
        run = gr.replay
        mode = "hipgraph"
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if out is None:
        out = fwd()
    ms = dt / args.steps * 1e3
    value = B * args.steps / dt
    tfl = value * QA32_MFLOP_PER_IMG_EVAL * 1e6 / 1e12
    res = {
        "metric": "eval images/sec (forward only) QA-ViT 32x32", "value": round(value, 1), "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"QAViT.py ({args.variant}) forward only, eval mode (BASELINE config 2)", "weights": f"qavit32_config(): img 32, patch 4, window 4, dilations (1,2), linformer_k 32; "
                   f"{sum(p.numel() for p in model.parameters()):,} parameters, random-init (key-seeded filler)", "global_batch": B, "per_gpu_batch": B, "image": "32x32x3", "parallelism": "dp1",
                   "launch": mode, "finite": bool(torch.isfinite(out.float()).all()), "algorithmic_tflops_per_s": round(tfl, 2), "algorithmic_mfma_frac": round(tfl / PEAK_BF16_TFLOPS, 5)},
    }
    if not args.no_kernel_timing:
        import importlib
        lib = importlib.import_module("qa-vit_amd.lib").load()
        fwd(); torch.cuda.synchronize()
        with KernelTimer(lib) as kt:
            fwd()
            fam = kt.summary()
        name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
        tfl_k = d["flops"] / (d["ms"] * 1e-3) / 1e12
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        peak_fl = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
        mfma_bound = (tfl_k / peak_fl) >= (gbs / PEAK_HBM_GBS)
        ach, peak, unit = (tfl_k, peak_fl, "TFLOP/s") if mfma_bound else (gbs, PEAK_HBM_GBS, "GB/s")
        res["roofline"] = {"bound": "mfma" if mfma_bound else "hbm", "kernel": name, "achieved": round(ach, 3), "peak": peak, "unit": unit, "frac": round(ach / peak, 5),
                           "traffic": None, "launches_per_step": d["launches"], "avg_launch_us": round(d["ms"] / d["launches"] * 1e3, 2),
                           "device_ms_per_step_all_entry_points": round(sum(v["ms"] for v in fam.values()), 3),
                           "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3)} for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:10]}}
    if not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline_q32(args.variant)
    print(json.dumps(res))


def main():
    args = parse()
    if args.config == "q32":
        if not args.eval:
            raise SystemExit("--config q32 is BASELINE config 2 (forward only): run it with --eval [--batch 512]")
        if "--batch" not in " ".join(sys.argv):
            args.batch = 512
        return bench_q32_eval(args)
    if args.eval:
        raise SystemExit("--eval is the q32 leg (BASELINE config 2); the c100 / tin legs time the full training step")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)                                   # does not return
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world > 1:
            args.gpus = world
        else:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: refusing to print a one-GPU number as n_gpus={args.gpus}")
    import qavit_amd as Q
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    from importlib import import_module
    par = import_module("qa-vit_amd.parallel")
    Q.lib.load()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    force_ddp = os.environ.get("QAVIT_FORCE_DDP", "0") != "0"      # exercise the data-parallel path on one rank
    if world > 1 or force_ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        par.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)      # flight recorder on: Trainer.capture drains the watchdog by it

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = Q.HQAViTConfig() if args.config == "c100" else Q.HQAViTTinyINConfig()
    model = Q.HQAViT(cfg)
    Q.fill_module(model)
    model = model.to(dev).train()
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, 3, cfg.img_size, cfg.img_size, generator=g).to(dev)
    y = torch.randint(0, cfg.num_classes, (B,), generator=g).to(dev)

    dp = None
    if world > 1 or force_ddp:
        tags = None
        if args.ddp_tags is not None:
            tags = () if args.ddp_tags == "none" else ("all" if args.ddp_tags == "all" else tuple(t for t in args.ddp_tags.split(",") if t))
        dp = par.DataParallel(model, sync_tags=tags)
    tcfg = Q.TrainingConfig(batch_size=B * world, use_amp=(cdt == torch.bfloat16), device_mix=args.mix)
    tr = Q.Trainer(model, tcfg, total_steps=100000, warmup_steps=1000, reducer=(dp.reducer if dp else None),
                   compute_dtype=cdt, order=par.bucket_order)
    if dp:
        dp.bind(tr)

    with_optim = not args.fwd_bwd_only
    mode = "eager"
    capture_error = None
    run = (lambda: tr.step(x, y)) if with_optim else (lambda: tr.fwd_bwd(x, y))
    if not args.no_graph:
        try:
            tr.capture(x, y, with_optim=with_optim, warmup=3)
            run = lambda: tr.replay()           # noqa: E731
            mode = "hipgraph"
        except Exception as e:                  # collectives / allocator that cannot be captured: stay eager, and SAY SO in the JSON line
            capture_error = f"{type(e).__name__}: {e}"
            if rank == 0:
                print(f"[bench] graph capture failed ({capture_error}); running eager", file=sys.stderr)
            torch.cuda.synchronize()
            tr.graph = None

    if rank == 0:
        print(f"[bench] mode={mode}, warm-up ...", file=sys.stderr, flush=True)
    for _ in range(args.warmup):
        run()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(tr.loss.item()) if mode == "hipgraph" else float(run().item())
    ms = dt / args.steps * 1e3
    value = B * world * args.steps / dt

    out = {
        "metric": "training images/sec (fwd+bwd) HQA-ViT CIFAR-100" if args.config == "c100" else "training images/sec (fwd+bwd) HQA-ViT Tiny-ImageNet", "value": round(value, 1), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": ("HQAViT_CIFAR100" if args.config == "c100" else "HQAViT_IN_Tiny") + " full training step (" + ("device CutMix/MixUp, " if args.mix else "") + "re-pack, fwd, CE loss, bwd, "
                               + ("grad all-reduce, " if world > 1 else "") + ("clip + fused AdamW)" if with_optim else "no optimiser)"),
                   "weights": f"{type(cfg).__name__}() defaults, {sum(p.numel() for p in model.parameters()):,} parameters, random-init (key-seeded filler)",
                   "global_batch": B * world, "per_gpu_batch": B, "image": f"{cfg.img_size}x{cfg.img_size}x3", "parallelism": f"dp{world}",
                   "launch": mode, "dropout": cfg.dropout, "drop_path": cfg.drop_path, "final_loss": round(loss_val, 4),
                   "ddp_sync_tags": (("all" if dp.sync_tags is None else list(dp.sync_tags)) if dp else None)},
    }
    step_tflops = value * (MFLOP_PER_IMG_TRAIN if args.config == "c100" else 6490.0) * 1e6 / 1e12     # 1,293 / 6,490 MFLOP per image
    out["config"]["algorithmic_tflops_per_s"] = round(step_tflops, 2)                     # the reference's own operation count
    out["config"]["algorithmic_mfma_frac"] = round(step_tflops / (PEAK_BF16_TFLOPS * world), 5)
    if args.config == "c100":
        need = value * NEEDED_MFLOP_PER_IMG_TRAIN * 1e6 / 1e12                             # what the path needs (skipped work not counted)
        out["config"]["needed_tflops_per_s"] = round(need, 2)
        out["config"]["needed_mfma_frac"] = round(need / (PEAK_BF16_TFLOPS * world), 5)
    if capture_error is not None:
        out["config"]["graph_capture_error"] = capture_error
    if getattr(tr, "watchdog_drain", None):
        out["config"]["watchdog_drain"] = tr.watchdog_drain

    if rank == 0 and world == 1 and not args.no_kernel_timing:
        # one instrumented eager step of the same workload: per-call HIP-event timing of every C-ABI entry point
        import importlib
        lib = importlib.import_module("qa-vit_amd.lib").load()
        for _ in range(2):
            tr.step(x, y) if with_optim else tr.fwd_bwd(x, y)
        torch.cuda.synchronize()
        with KernelTimer(lib) as kt:
            tr.step(x, y) if with_optim else tr.fwd_bwd(x, y)
            fam = kt.summary()
        name, d = max(fam.items(), key=lambda kv: kv[1]["ms"])
        avg_ms = d["ms"] / d["launches"]
        tfl = d["flops"] / (d["ms"] * 1e-3) / 1e12
        gbs = d["bytes"] / (d["ms"] * 1e-3) / 1e9
        peak_fl = PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3
        # which roof is nearer for this entry point's work mix
        mfma_bound = (tfl / peak_fl) >= (gbs / PEAK_HBM_GBS)
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        # the committed counter passes were taken on the default workload (C100, B = 1024, bf16): null for anything else
        default_workload = args.config == "c100" and args.batch == 1024 and args.dtype == "bf16"
        if os.path.exists(tp) and default_workload:
            try:
                traffic = json.load(open(tp)).get(name)
                if isinstance(traffic, dict):            # HBM bytes per launch from the rocprofv3 PMC passes (tools/pmc_traffic.py)
                    traffic = traffic.get("hbm_bytes_per_kernel")
            except Exception:
                traffic = None
        # MFMA utilisation from hardware counters (rocprofv3 --pmc pass folded by tools/pmc_mfma.py into profiles/mfma_util.json)
        mfma_util = None
        mp = os.path.join(ROOT, "profiles", "mfma_util.json")
        if os.path.exists(mp) and default_workload:
            try:
                mj = json.load(open(mp))
                mfma_util = {k: v["mfma_util_pct"] for k, v in mj.items() if isinstance(v, dict) and "mfma_util_pct" in v}
            except Exception:
                mfma_util = None
        ach, peak, unit = (tfl, peak_fl, "TFLOP/s") if mfma_bound else (gbs, PEAK_HBM_GBS, "GB/s")
        # The same fraction from the TRACKED rocprofv3 summary (profiles/rNN_family_summary.txt = tools/kfamily.py over --kernel-trace --stats of
        # this command): this run's algorithmic bytes / flops of the family over the profiled family time, so the line can be re-derived from
        # files in the repository alone.  Under the profiler every kernel reads a little longer than between the event brackets above.
        from_profiles = None
        if default_workload:
            import glob
            import re
            fams = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_family_summary.txt")))
            if fams:
                try:
                    for line in open(fams[-1]):
                        m = re.match(r"^(\S.*?)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)\s+([0-9.]+)%\s*$", line)
                        if m and m.group(1).strip() == name:
                            fam_ms = float(m.group(3))
                            p_ach = (d["flops"] / 1e12 if mfma_bound else d["bytes"] / 1e9) / (fam_ms * 1e-3)
                            from_profiles = {"file": os.path.relpath(fams[-1], ROOT), "family": name, "kernels_per_step": float(m.group(2)),
                                             "family_ms_per_step": fam_ms, "achieved": round(p_ach, 3), "frac": round(p_ach / peak, 5)}
                except Exception:
                    from_profiles = None
        out["roofline"] = {"bound": "mfma" if mfma_bound else "hbm", "kernel": name, "achieved": round(ach, 3), "peak": peak,
                           "unit": unit, "frac": round(ach / peak, 5), "traffic": traffic,
                           "launches_per_step": d["launches"], "avg_launch_us": round(avg_ms * 1e3, 2),
                           "flops_per_launch": round(d["flops"] / d["launches"]), "bytes_per_launch": round(d["bytes"] / d["launches"]),
                           "tflops": round(tfl, 2), "gbytes_per_s": round(gbs, 1), "event_bracket_overhead_us": round(kt.empty_ms * 1e3, 2),
                           "device_ms_per_step_all_entry_points": round(sum(v["ms"] for v in fam.values()), 3),
                           "mfma_util_pct_by_family_pmc": mfma_util,
                           "from_profiles": from_profiles,
                           "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                            "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2),
                                            "gbs": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1)}
                                        for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("QAVIT_BENCH_FAMILIES", "12"))]}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(tin=args.config == "tin")
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
