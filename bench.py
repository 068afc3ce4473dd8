#!/usr/bin/env python3
"""Benchmark of the hot path BASELINE.json names: HQA-ViT CIFAR-100 training step on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

A "step" is one full training step of HQAViT(HQAViTConfig()) on a synthetic 32x32x3 batch of 1024 images per
GPU (BASELINE.json configs[2]/[3]): weight re-pack, forward, loss, backward, gradient all-reduce (N > 1),
clipping and fused AdamW -- nothing is skipped inside the timed region.  bf16 activations, fp32 accumulate
and master weights; reference-default dropout / drop-path.  Inputs are resident in HBM before timing starts.

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

MFLOP_PER_IMG_TRAIN = 1293.0       # fwd+bwd algorithmic work, HQA-ViT CIFAR-100
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=1024, help="images per GPU")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of one hipGraph per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--fwd-bwd-only", action="store_true", help="time forward+backward(+all-reduce) without the optimiser")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------
# per-launch kernel timing with HIP events (roofline)
# ---------------------------------------------------------------------------------------------------
class KernelTimer:
    """Wraps the raw kernel wrappers of one eager step with torch.cuda.Event pairs recorded on the launch stream
    (kernels are enqueued on torch's current stream, so the events bracket exactly one launch)."""

    def __init__(self, K):
        self.K, self.rec, self._orig = K, [], {}

    def _wrap(self, name, flops_fn):
        orig = getattr(self.K, name)
        self._orig[name] = orig

        def f(*a, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*a, **kw)
            e1.record()
            self.rec.append((name, flops_fn(*a, **kw), e0, e1))
            return r
        setattr(self.K, name, f)

    def __enter__(self):
        self._wrap("gemm_nt", lambda A, B, C, M, N, Kd, *r, **kw: 2.0 * M * N * Kd)
        self._wrap("gemm_tn", lambda A, B, C, M, N, Kd, *r, **kw: 2.0 * M * N * Kd)

        def attn_flops(a):
            nk = (a.KC if a.mode == 0 else a.L) + a.S
            f = 4.0 * a.G * a.H * a.Nq * nk * a.D
            if a.mode == 0:
                f += 4.0 * a.G * a.H * a.KC * a.L * a.D
            return f
        self._wrap("attn_fwd", attn_flops)
        self._wrap("attn_bwd", lambda a: 2.5 * attn_flops(a))
        return self

    def __exit__(self, *exc):
        for n, o in self._orig.items():
            setattr(self.K, n, o)

    def summary(self):
        torch.cuda.synchronize()
        fam = {}
        for name, fl, e0, e1 in self.rec:
            d = fam.setdefault(name, dict(launches=0, ms=0.0, flops=0.0))
            d["launches"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
        return fam


def cpu_baseline(batch=32, budget_s=20.0):
    """CPU oracle (a port of the reference's PyTorch-CPU path) timed on this host: HQA-ViT C100 train step."""
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
    cfg = Q.HQAViTConfig()
    model = Q.HQAViT(cfg)
    Q.fill_module(model)
    P = {k: v.clone() for k, v in model.state_dict().items()}
    for k in list(P):
        if k.endswith(("global_bank.global_k", "global_bank.global_v", "global_bank.update_count")):
            P[k] = P["global_bank." + k.rsplit(".", 1)[-1]]
    for n, _ in model.named_parameters():
        P[n].requires_grad_(True)
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(batch, 3, 32, 32, generator=g)
    y = torch.randint(0, 100, (batch,), generator=g)
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
            "sample": f"HQA-ViT CIFAR-100 train fwd+bwd, B={batch}, fp32, {len(times)} timed steps after 2 warm-up, median {med * 1e3:.0f} ms/step"}


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        args.gpus = world
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
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    cdt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    cfg = Q.HQAViTConfig()
    model = Q.HQAViT(cfg)
    Q.fill_module(model)
    model = model.to(dev).train()
    B = args.batch
    g = torch.Generator().manual_seed(1234 + rank)
    x = torch.randn(B, 3, 32, 32, generator=g).to(dev)
    y = torch.randint(0, cfg.num_classes, (B,), generator=g).to(dev)

    dp = None
    if world > 1 or force_ddp:
        dp = par.DataParallel(model)
    tcfg = Q.TrainingConfig(batch_size=B * world, use_amp=(cdt == torch.bfloat16))
    tr = Q.Trainer(model, tcfg, total_steps=100000, warmup_steps=1000, reducer=(dp.reducer if dp else None),
                   compute_dtype=cdt, order=par.bucket_order)
    if dp:
        dp.bind(tr)

    with_optim = not args.fwd_bwd_only
    mode = "eager"
    run = (lambda: tr.step(x, y)) if with_optim else (lambda: tr.fwd_bwd(x, y))
    if not args.no_graph:
        try:
            tr.capture(x, y, with_optim=with_optim, warmup=3)
            run = lambda: tr.replay()           # noqa: E731
            mode = "hipgraph"
        except Exception as e:                  # collectives / allocator that cannot be captured: stay eager
            if rank == 0:
                print(f"[bench] graph capture failed ({type(e).__name__}: {e}); running eager", file=sys.stderr)
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
        "metric": "training images/sec (fwd+bwd) HQA-ViT CIFAR-100", "value": round(value, 1), "unit": "images/sec",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "HQAViT_CIFAR100 full training step (re-pack, fwd, CE loss, bwd, "
                               + ("grad all-reduce, " if world > 1 else "") + ("clip + fused AdamW)" if with_optim else "no optimiser)"),
                   "model": "HQAViT(HQAViTConfig()) 6,472,037 params, random-init (key-seeded filler)",
                   "global_batch": B * world, "per_gpu_batch": B, "image": "32x32x3", "parallelism": f"dp{world}",
                   "launch": mode, "dropout": cfg.dropout, "drop_path": cfg.drop_path, "final_loss": round(loss_val, 4)},
    }
    step_tflops = value * MFLOP_PER_IMG_TRAIN * 1e6 / 1e12
    out["config"]["model_tflops_per_s"] = round(step_tflops, 2)
    out["config"]["model_mfma_frac"] = round(step_tflops / (PEAK_BF16_TFLOPS * world), 5)

    if rank == 0 and world == 1 and not args.no_kernel_timing:
        # one instrumented eager step of the same workload: per-launch HIP-event timing of the kernel families
        for _ in range(2):
            tr.step(x, y) if with_optim else tr.fwd_bwd(x, y)
        torch.cuda.synchronize()
        with KernelTimer(K) as kt:
            tr.step(x, y) if with_optim else tr.fwd_bwd(x, y)
            fam = kt.summary()
        dom = max(fam.items(), key=lambda kv: kv[1]["ms"])
        name, d = dom
        avg_ms = d["ms"] / d["launches"]
        ach = d["flops"] / d["launches"] / (avg_ms * 1e-3) / 1e12
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get(name)
            except Exception:
                traffic = None
        out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 3), "peak": PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3,
                           "unit": "TFLOP/s", "frac": round(ach / (PEAK_BF16_TFLOPS if args.dtype == "bf16" else 157.3), 5),
                           "traffic": traffic, "launches_per_step": d["launches"], "avg_launch_us": round(avg_ms * 1e3, 2),
                           "flops_per_launch": round(d["flops"] / d["launches"]),
                           "families": {k: {"launches": v["launches"], "ms": round(v["ms"], 3),
                                            "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 2)} for k, v in fam.items()}}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
