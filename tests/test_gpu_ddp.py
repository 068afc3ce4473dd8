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


def _worker(rank, world, port, out):
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
    dp = par.DataParallel(model)
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
    res = dict(losses=losses, gnorm=tr.grad_norm(), n_buckets=len(dp.reducer.bounds),
               p_sum=float(tr.flat_p.double().sum()), p_abs=float(tr.flat_p.double().abs().sum()),
               bank=float(model.global_bank.global_k.double().abs().sum()), count=int(model.global_bank.update_count),
               g_abs=float(tr.flat_g.double().abs().sum()))
    torch.save(res, os.path.join(out, f"r{rank}.pt"))
    dist.destroy_process_group()


def test_two_rank_step_keeps_replicas_identical(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert r0["n_buckets"] >= 3
    # replicas: identical parameters, averaged gradients, bank and counter after two optimiser steps
    for k in ("p_sum", "p_abs", "bank", "g_abs", "gnorm"):
        assert abs(r0[k] - r1[k]) <= 1e-6 * max(abs(r0[k]), 1e-12), (k, r0[k], r1[k])
    assert r0["count"] == r1["count"] == 48
    assert all(l == l and 0 < l < 20 for l in r0["losses"] + r1["losses"])
    assert r0["losses"] != r1["losses"]                     # different shards
