"""Data-parallel training step of the REAL model on the GPU with two ranks.  A one-GPU box cannot give RCCL two devices,
so both ranks share cuda:0 and exchange through gloo (which stages CUDA tensors through the host): the collectives are
slow but the code under test -- DataParallel (broadcast, bank-statistics all-reduce), SyncPoints, the bucketed
GradReducer over the Trainer's flat gradient buffer, the fused optimiser -- is exactly what runs over RCCL."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, tags):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    import qavit_amd as Q
    par = import_module("qa-vit_amd.parallel")
    torch.cuda.set_device(0)
    cfg = Q.HQAViTConfig(dropout=0.0, drop_path=0.0)
    torch.manual_seed(50 + rank)                          # ranks start different: the constructor's broadcast must fix it
    model = Q.HQAViT(cfg).cuda().train()
    if rank == 0:
        Q.fill_module(model)
    for m in model.modules():                             # SplitFusion.cat_mlp hard-wires Dropout(0.1) (HQAViT_CIFAR100.py:930)
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    # sync points: buckets reduced during backward (each one flushes the weight-gradient problems of ITS address range only:
    # DeferDW.flush_range) and after it; "all" = the seven stage boundaries, () = one reduction after backward, None = the default
    dp = par.DataParallel(model, sync_tags=tags)
    assert dp.sync_tags == (("fuse3",) if tags is None else None if tags == "all" else tuple(tags))
    tr = Q.Trainer(model, Q.TrainingConfig(batch_size=16, use_amp=False), total_steps=100, warmup_steps=10,
                   reducer=dp.reducer, compute_dtype=torch.float32, order=par.bucket_order)
    dp.bind(tr)
    g = torch.Generator().manual_seed(11)
    X = torch.randn(16, 3, 32, 32, generator=g)
    Y = torch.randint(0, 100, (16,), generator=g)
    x, y = X[rank * 8:(rank + 1) * 8].cuda(), Y[rank * 8:(rank + 1) * 8].cuda()
    losses = []
    for _ in range(2):
        losses.append(float(tr.step(x, y)))
        dp.after_step()
    torch.cuda.synchronize()
    probe = {n: p.detach().reshape(-1)[:64].cpu().clone() for n, p in model.named_parameters()
             if n in ("head.weight", "cnn_stem.stem.1.weight", "cnn_stem.stage3.1.bias", "stage1_blocks.0.quad_block.swa.qkv.weight", "global_bank.global_k")}
    res = dict(losses=losses, gnorm=tr.grad_norm(), n_buckets=len(dp.reducer.bounds), probe=probe,
               bn_mean=model.cnn_stem.stem[1].running_mean.cpu().clone(), bn_var=model.cnn_stem.stage1[1].running_var.cpu().clone(),
               flat_g=tr.flat_g.cpu().clone(), names=tr.names, offsets=tr.offsets,
               p_sum=float(tr.flat_p.double().sum()), p_abs=float(tr.flat_p.double().abs().sum()),
               bank=float(model.global_bank.global_k.detach().double().abs().sum()), count=int(model.global_bank.update_count),
               g_abs=float(tr.flat_g.double().abs().sum()))
    torch.save(res, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def _single_process_reference():
    """The same two optimiser steps in ONE process on the concatenated B = 16 batch (what the reference's single-GPU loop
    computes): the target the exact data-parallel modes (bank statistics, SyncBN) must reproduce."""
    sys.path.insert(0, ROOT)
    import qavit_amd as Q
    cfg = Q.HQAViTConfig(dropout=0.0, drop_path=0.0)
    model = Q.HQAViT(cfg).cuda().train()
    Q.fill_module(model)
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    tr = Q.Trainer(model, Q.TrainingConfig(batch_size=16, use_amp=False), total_steps=100, warmup_steps=10, compute_dtype=torch.float32)
    g = torch.Generator().manual_seed(11)
    X = torch.randn(16, 3, 32, 32, generator=g).cuda()
    Y = torch.randint(0, 100, (16,), generator=g).cuda()
    losses = [float(tr.step(X, Y)) for _ in range(2)]
    torch.cuda.synchronize()
    return model, tr, losses


@pytest.mark.parametrize("tags", [("fuse3", "stage1_blocks"), "all", (), None], ids=["two_sync_points", "all_seven", "none", "default"])
def test_two_rank_step_keeps_replicas_identical(tmp_path, tags):
    """Whatever the sync points -- two, all seven (every one a partial weight-gradient flush by address range), none, or the default
    for world > 1 (one overlapped bucket prefix) -- the reduced gradients and the parameters after two steps are those of the
    single-process B = 16 step."""
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), tags), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt", weights_only=False), torch.load(tmp_path / "r1.pt", weights_only=False)
    # ---- 2 ranks x 8 images == one process x 16 images (exact bank statistics + SyncBN): parameters after two optimiser
    # steps, the last step's averaged gradient, the bank, BatchNorm running statistics, the global gradient norm
    model, tr, losses = _single_process_reference()
    def rel(a, b):
        return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
    assert abs(0.5 * (r0["losses"][0] + r1["losses"][0]) - losses[0]) <= 1e-5 * abs(losses[0])       # step 0: identical weights
    assert abs(0.5 * (r0["losses"][1] + r1["losses"][1]) - losses[1]) <= 1e-4 * abs(losses[1])
    params = dict(model.named_parameters())
    for n, v in r0["probe"].items():
        assert rel(v, params[n].detach().reshape(-1)[:64].cpu()) <= 1e-4, n
    assert rel(r0["bn_mean"], model.cnn_stem.stem[1].running_mean.cpu()) <= 1e-5
    assert rel(r0["bn_var"], model.cnn_stem.stage1[1].running_var.cpu()) <= 1e-5
    assert abs(r0["gnorm"] - tr.grad_norm()) <= 1e-4 * tr.grad_norm()
    ref_g = {n: tr.flat_g[o:o + p.numel()].cpu() for n, o, p in zip(tr.names, tr.offsets, tr.params)}
    gmax = max(float(v.abs().max()) for v in ref_g.values())
    worst = []
    from conftest import zero_by_construction
    for n, o in zip(r0["names"], r0["offsets"]):
        if zero_by_construction(n):                         # exactly-zero gradients: what is there is round-off
            continue
        g0 = r0["flat_g"][o:o + ref_g[n].numel()]
        worst.append((float((g0 - ref_g[n]).abs().max()) / max(float(ref_g[n].abs().max()), 1e-3 * gmax), n))
    worst.sort(reverse=True)
    assert worst[0][0] <= 2e-3, worst[:5]
    assert r0["n_buckets"] >= 3
    # replicas: identical parameters, averaged gradients, bank and counter after two optimiser steps
    for k in ("p_sum", "p_abs", "bank", "g_abs", "gnorm"):
        assert abs(r0[k] - r1[k]) <= 1e-6 * max(abs(r0[k]), 1e-12), (k, r0[k], r1[k])
    assert r0["count"] == r1["count"] == 48
    assert all(l == l and 0 < l < 20 for l in r0["losses"] + r1["losses"])
    assert r0["losses"] != r1["losses"]                     # different shards
