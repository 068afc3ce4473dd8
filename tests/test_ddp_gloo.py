"""Data-parallel logic on CPU with the gloo backend (world_size 2): gradient reducer (buckets, sync points,
averaging), parameter broadcast, and the exact-mode bank statistics exchange.  The HIP model cannot run on CPU,
so a small pure-torch stand-in with the same naming scheme drives the SAME DataParallel / GradReducer code."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Toy(torch.nn.Module):
    """Names follow the real model (stageK_blocks / fuseK / head / pos_embed) so bucket_order and tags apply."""

    def __init__(self):
        super().__init__()
        d = 24
        self.pos_embed = torch.nn.Parameter(torch.zeros(1, 4, d))
        for i in (1, 2, 3, 4):
            setattr(self, f"stage{i}_blocks", torch.nn.ModuleList([torch.nn.Linear(d, d), torch.nn.Linear(d, d)]))
        for i in (2, 3, 4):
            setattr(self, f"fuse{i}", torch.nn.Linear(d, d))
        self.norm = torch.nn.LayerNorm(d)
        self.head = torch.nn.Linear(d, 5)
        self._sync_reducer = None

    def _sync(self, t, tag):
        if self._sync_reducer is not None and torch.is_grad_enabled():
            from importlib import import_module
            return import_module("qa-vit_amd.parallel").SyncPoint.apply(t, self._sync_reducer, tag)
        return t

    def forward(self, x):
        t = x + self.pos_embed
        for i in (1, 2, 3, 4):
            if i >= 2:
                t = self._sync(t, f"fuse{i}")
                t = torch.tanh(getattr(self, f"fuse{i}")(t))
            t = self._sync(t, f"stage{i}_blocks")
            for blk in getattr(self, f"stage{i}_blocks"):
                t = t + torch.tanh(blk(t))
        return self.head(self.norm(t).mean(1))


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from importlib import import_module
    import qavit_amd  # noqa: F401
    par = import_module("qa-vit_amd.parallel")
    torch.manual_seed(100 + rank)                      # different init per rank: broadcast must fix it
    model = Toy()
    dp = par.DataParallel(model, bucket_bytes=4096, sync_tags="all")     # tiny buckets -> several per stage, reduced at every sync point of backward
    named = par.bucket_order(list(model.named_parameters()))
    names = [n for n, _ in named]
    offs = [0]
    for _, p in named:
        offs.append(offs[-1] + p.numel())
    flat_g = torch.zeros(offs[-1])
    for (n, p), o in zip(named, offs):
        p.grad = flat_g[o:o + p.numel()].view_as(p)

    class T:
        pass
    tr = T()
    tr.names, tr.offsets, tr.flat_g = names, offs, flat_g
    dp.bind(tr)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(8, 4, 24, generator=g)
    Y = torch.randint(0, 5, (8,), generator=g)
    xs, ys = X[rank * 4:(rank + 1) * 4], Y[rank * 4:(rank + 1) * 4]
    dp.reducer.begin_step()
    loss = torch.nn.functional.cross_entropy(model(xs), ys)
    loss.backward()
    launched_before_finish = dp.reducer._next
    dp.reducer.finish(flat_g)
    # single-process reference on the full batch with rank 0's weights (= the broadcast weights)
    ref = Toy()
    ref.load_state_dict(model.state_dict())
    torch.nn.functional.cross_entropy(ref(X), Y).backward()
    err = max(float((p.grad - dict(ref.named_parameters())[n].grad).abs().max()) for n, p in model.named_parameters())
    # bank statistics exchange (exact mode): SUM over ranks, returns the global batch
    acc = torch.full((6,), float(rank + 1))
    total = dp._bank_all_reduce(acc, 4)
    w0 = model.head.weight.detach().clone()
    dist.broadcast(w0, src=0)
    same = bool(torch.equal(w0, model.head.weight.detach()))
    if rank == 0:
        torch.save(dict(err=err, nb=len(dp.reducer.bounds), early=launched_before_finish, acc=acc, total=total, same=same), out)
    dist.destroy_process_group()


def test_two_rank_gradient_allreduce_and_bank_sync(tmp_path):
    out = str(tmp_path / "r.pt")
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r["err"] <= 1e-6, r            # averaged sharded gradients == full-batch gradients
    assert r["nb"] >= 4                   # the flat buffer really was cut into several buckets
    assert r["early"] >= 1                # and some of them were launched from SyncPoints during backward
    assert torch.equal(r["acc"], torch.full((6,), 3.0)) and r["total"] == 8
    assert r["same"]
