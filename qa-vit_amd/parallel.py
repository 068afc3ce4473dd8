"""Data parallelism for the training path -- NEW functionality (the reference has no distributed code at all,
SURVEY.md section 2): one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI on ROCm,
"gloo" on CPU for tests).

Design (SURVEY.md sections 5 and 8e):
  * the global batch is sharded, weights replicated; the one exchange step is a SUM all-reduce of the flat
    fp32 gradient buffer, cut into contiguous buckets ordered by backward completion (head and last stage
    first, CNN stem / bank / embeddings last).  ``SyncPoint`` autograd nodes placed between stages fire during
    backward and launch the buckets whose gradients are complete on a side stream, overlapping the remaining
    backward; ``finish()`` launches the rest, waits, and divides by the world size.  25.7 MB of gradients
    are tiny against 7 x 153 GB/s of xGMI per GPU, so a handful of >= 4 MB buckets keeps every collective
    bandwidth- rather than latency-bound.
  * exception 1, ``GlobalTokenBank.write``: its batch mean spans the GLOBAL batch, so the [S,C] statistics
    are all-reduced (SUM) before the clamp/update -- 24 (C100) sequential 12 KB collectives per forward,
    latency-bound; exact mode is the default, ``bank_sync="local"`` keeps per-rank banks and re-broadcasts
    rank 0's bank every ``bank_broadcast_every`` steps.
  * exception 2, BatchNorm in the CNN stem (4 layers): ``bn_sync="exact"`` (default) all-reduces the per-channel column
    sums between the statistics pass and the apply pass of csrc/bnorm.hip, forward and backward (SyncBN: 2 x 4 small
    collectives per step), so an N-rank step equals the single-process step on the global batch; ``bn_sync="local"``
    keeps per-rank statistics (stock DDP semantics).  Buffers are broadcast from rank 0 at construction.
  * dropout / drop-path RNG streams differ per rank (seed + rank).
"""
import os
import pickle
import time
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_process_group(backend: str = "nccl", **kw):
    """``dist.init_process_group`` with the process group's flight recorder ON (TORCH_FR_BUFFER_SIZE, read when the NCCL / RCCL group
    is created): ``drain_watchdog`` reads from it which collectives the watchdog thread has retired."""
    os.environ.setdefault("TORCH_FR_BUFFER_SIZE", "2000")            # (older builds read TORCH_NCCL_TRACE_BUFFER_SIZE)
    if "TORCH_NCCL_TRACE_BUFFER_SIZE" not in os.environ and not hasattr(torch._C._distributed_c10d, "_dump_fr_trace"):
        os.environ["TORCH_NCCL_TRACE_BUFFER_SIZE"] = "2000"
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # RCCL over dmabuf IPC
    return dist.init_process_group(backend, **kw)


def _active_collectives() -> Optional[int]:
    """Number of collectives still on the watchdog thread's list, from the flight recorder: entries whose ``retired`` flag the watchdog
    has not set yet.  (``state == completed`` is NOT that: the dump itself marks an entry completed as soon as its end event has
    fired, while the watchdog keeps polling the work until its own next pass -- measured on MI355X: completed at once, retired 81 ms
    later, tools/watchdog_probe.py.)  None when the recorder is off or this build has none: nothing can be observed."""
    try:
        from torch._C import _distributed_c10d as c10d
        entries = pickle.loads(c10d._dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False)).get("entries")
        if not entries or "retired" not in entries[0]:
            return None                                     # recorder off (or no collective was ever issued: nothing to drain either)
        return sum(0 if e.get("retired") else 1 for e in entries)
    except Exception:
        return None


def drain_watchdog(timeout_s: float = 20.0, poll_s: float = 0.02) -> str:
    """Block until the process group's watchdog thread has RETIRED every collective issued so far.  Call after a device
    synchronize, before hipGraph capture: the watchdog polls the end events of the collectives on its list (hipEventQuery, one pass
    per 100 ms) until it has seen each complete; if the stream such an event was recorded on joins a capture meanwhile, HIP answers
    the query with hipErrorCapturedEvent and the watchdog takes the process down.  Round 3 slept 0.5 s here and hoped; this reads the
    list's state from the flight recorder and waits for it to be empty -- or raises after ``timeout_s``.  Returns how it decided."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_backend() != "nccl":
        return "no nccl group"
    n = _active_collectives()
    if n is None:
        # nothing observable: the one fallback left is the watchdog's own period -- several passes, and say so
        time.sleep(1.0)
        return "flight recorder off: slept 1.0 s (start the group with parallel.init_process_group to make this deterministic)"
    t0 = time.monotonic()
    while n:
        if time.monotonic() - t0 > timeout_s:
            raise RuntimeError(f"drain_watchdog: {n} collective(s) still on the watchdog's list after {timeout_s} s; refusing to capture")
        time.sleep(poll_s)
        n = _active_collectives()
        if n is None:
            raise RuntimeError("drain_watchdog: the flight recorder stopped answering")
    return f"watchdog list empty after {time.monotonic() - t0:.3f} s"


def bucket_order(named: Sequence[Tuple[str, torch.nn.Parameter]]):
    """Order parameters by when their gradient completes in backward: head/norm, stage4, fuse4, stage3, ...,
    stage1, then everything that stays live until the very end (embeddings, bank, CNN lateral path)."""
    def key(name: str) -> int:
        if name.startswith(("head.", "norm.")):
            return 0
        for i, tag in enumerate(("stage4_blocks", "fuse4", "stage3_blocks", "fuse3", "stage2_blocks", "fuse2", "stage1_blocks")):
            if name.startswith(tag):
                return 1 + i
        if name.startswith("blocks."):                      # QAViT: blocks.<i>, later blocks finish first
            try:
                return 1 + (1000 - int(name.split(".")[1]))
            except ValueError:
                return 1
        return 10_000
    return sorted(named, key=lambda kv: key(kv[0]))


class SyncPoint(torch.autograd.Function):
    """Identity in forward; in backward tells the reducer that every parameter used AFTER this point in the
    forward pass has its complete gradient."""

    @staticmethod
    def forward(ctx, x, reducer, tag):
        ctx.reducer, ctx.tag = reducer, tag
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        rng = ctx.reducer.ready_range(ctx.tag)
        if rng is not None:                    # a bucket becomes complete here: its deferred weight-gradient GEMMs must land before it is reduced
            try:
                from .functional import DeferDW
                if GradReducer.FLUSH_ALL:
                    DeferDW.flush()
                else:
                    DeferDW.flush_range(*rng)
            except ImportError:
                pass
            ctx.reducer.reached(ctx.tag)
        return g, None, None


class GradReducer:
    """Bucketed all-reduce of a flat gradient buffer, overlapped with backward on a side stream."""

    # QAVIT_DDP_FLUSH_ALL=1: the round-2 behaviour (every sync point launches everything queued, the lateral stream's problems included)
    FLUSH_ALL = __import__("os").environ.get("QAVIT_DDP_FLUSH_ALL", "0") != "0"

    def __init__(self, group=None, bucket_bytes: int = 4 << 20):
        self.group = group
        self.world = dist.get_world_size(group)
        self.bucket_bytes = bucket_bytes
        self.bounds: List[Tuple[int, int]] = []       # element ranges, in launch order
        self.ready_at = {}                            # tag -> number of buckets complete when the tag fires
        self._next = 0
        self._flat = None
        self._side = None
        self._works = []

    def plan(self, names: Sequence[str], offsets: Sequence[int], tags: Sequence[Tuple[str, Callable[[str], bool]]]):
        """``names/offsets`` describe the flat buffer (already in bucket_order).  ``tags`` lists, in backward
        firing order, (tag, predicate(name) -> "complete once this tag has fired")."""
        total = offsets[-1]
        per = max(1, self.bucket_bytes // 4)
        # tag boundaries in elements: parameters complete at tag k form a prefix of the flat buffer
        done_upto = []
        pos = 0
        for tag, pred in tags:
            while pos < len(names) and pred(names[pos]):
                pos += 1
            done_upto.append((tag, offsets[pos]))
        # cut buckets: never across a tag boundary's "ready" prefix unless the piece is small
        cuts = sorted({0, total, *[e for _, e in done_upto]})
        bounds = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            s = a
            while s < b:
                e = min(b, s + per)
                if b - e < per // 4:
                    e = b
                bounds.append((s, e))
                s = e
        merged = []
        for s, e in bounds:                            # merge slivers into their predecessor
            if merged and (e - s) < per // 16:
                merged[-1] = (merged[-1][0], e)
            else:
                merged.append((s, e))
        self.bounds = merged
        self.ready_at = {}
        for tag, upto in done_upto:
            self.ready_at[tag] = sum(1 for (_, e) in merged if e <= upto)
        return merged

    def attach(self, flat: torch.Tensor):
        self._flat = flat
        if flat.is_cuda:
            self._side = torch.cuda.Stream(device=flat.device)

    def begin_step(self):
        self._next = 0
        self._works = []

    def _launch(self, upto_bucket: int):
        if self._flat is None or upto_bucket <= self._next:
            return
        flat = self._flat
        if self._side is not None:
            self._side.wait_stream(torch.cuda.current_stream(flat.device))
            with torch.cuda.stream(self._side):
                for s, e in self.bounds[self._next:upto_bucket]:
                    dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM, group=self.group)
        else:
            for s, e in self.bounds[self._next:upto_bucket]:
                self._works.append(dist.all_reduce(flat[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._next = upto_bucket

    def ready_range(self, tag):
        """Address range [lo, hi) of the flat gradient buffer that becomes reducible when ``tag`` fires (the buckets not yet launched),
        or None when no new bucket is complete at this tag."""
        upto = self.ready_at.get(tag, 0)
        if self._flat is None or upto <= self._next:
            return None
        base = self._flat.data_ptr()
        return base + 4 * self.bounds[self._next][0], base + 4 * self.bounds[upto - 1][1]

    def reached(self, tag):
        self._launch(self.ready_at.get(tag, 0))

    def finish(self, flat: torch.Tensor):
        if self._flat is None:
            self.attach(flat)
        self._launch(len(self.bounds))
        if self._side is not None:
            torch.cuda.current_stream(flat.device).wait_stream(self._side)
        for w in self._works:
            w.wait()
        self._works = []
        flat.mul_(1.0 / self.world)


class DataParallel:
    """Wires a model + Trainer for data-parallel training.

        dp = DataParallel(model)                       # broadcast weights/buffers, per-rank RNG, bank hook
        trainer = Trainer(model, cfg, steps, reducer=dp.reducer, order=bucket_order)
        dp.bind(trainer)                               # plan buckets over the trainer's flat gradient buffer
    """

    def __init__(self, model: torch.nn.Module, group=None, bucket_bytes: int = 4 << 20, bank_sync: str = "exact",
                 bank_broadcast_every: int = 50, seed: int = 0x5EED, bn_sync: str = "exact", sync_tags=None):
        """``sync_tags``: the stage boundaries at which complete buckets are all-reduced DURING backward (a tuple of tags, "all" = all
        seven, () = none: one reduction after backward; the default depends on the world size, see below; the environment variable
        QAVIT_DDP_TAGS = comma-separated tags or "all" overrides).  Every sync point costs a grouped weight-gradient launch over fewer problems than the
        single end-of-backward one, so on a fast fabric fewer, larger overlapped reductions win."""
        if not dist.is_initialized():
            raise RuntimeError("DataParallel needs torch.distributed.init_process_group first")
        self.model, self.group = model, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.bank_sync, self.bank_every, self._steps = bank_sync, bank_broadcast_every, 0
        env = __import__("os").environ.get("QAVIT_DDP_TAGS")
        if env is not None:
            self.sync_tags = None if env == "all" else tuple(t for t in env.split(",") if t)
        elif sync_tags is not None:
            self.sync_tags = None if sync_tags == "all" else tuple(sync_tags)
        else:
            # default by world size.  ONE rank: no sync point -- there is nothing to overlap (RCCL launches no kernel for a one-rank collective)
            # and every sync point splits the single end-of-backward weight-gradient launch and puts a flush on backward's critical path
            # (measured on one MI355X, B = 1024, one rank over RCCL, captured; DESIGN.md section 6: no sync point 10.26 ms / step = the step
            # without any data-parallel machinery, one sync point (fuse3) 10.40, all seven 10.61).  MORE than one rank: one sync point in the
            # middle of backward ("fuse3"): the bucket prefix head .. fuse3 (about half of the 25.7 MB) is all-reduced on the side stream
            # while stages 2 and 1 and the lateral path run their backward -- the north star's "overlapped with the backward pass".  Whether
            # an exposed all-reduce is cheaper than the split flush on 8 GPUs has not been measured (no multi-GPU node in this pipeline), so
            # the overlapped form stays the default there; sync_tags=() / "all", QAVIT_DDP_TAGS or bench.py --ddp-tags select the others.
            self.sync_tags = ("fuse3",) if self.world > 1 else ()
        with torch.no_grad():
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, src=0, group=group)
        self.reducer = GradReducer(group, bucket_bytes)
        dev = next(model.parameters()).device
        if dev.type == "cuda":
            from . import kernels as K
            K.Runtime.get(dev).seed(seed + self.rank)
        if hasattr(model, "set_bank_sync"):
            model.set_bank_sync(self._bank_all_reduce if bank_sync == "exact" else None)
        if hasattr(model, "_sync_reducer"):
            model._sync_reducer = self.reducer
        if bn_sync not in ("exact", "local"):
            raise ValueError("bn_sync must be 'exact' or 'local'")
        from .functional import BatchNormFn
        BatchNormFn.sync = self._bn_all_reduce if (bn_sync == "exact" and self.world > 1) else None

    def _bn_all_reduce(self, stats: torch.Tensor) -> int:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=self.group)
        return self.world

    def _bank_all_reduce(self, acc: torch.Tensor, local_batch: int) -> int:
        dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=self.group)
        return local_batch * self.world

    def bind(self, trainer):
        names, offsets = trainer.names, trainer.offsets
        stage_tags = []
        if hasattr(self.model, "blocks"):                  # QAViT: sync points before blocks 2, 4, 6, ... (models.QAViT.forward)
            nblk = len(self.model.blocks)
            for i in range(nblk - 1, 0, -1):
                if i % 2 == 0:
                    stage_tags.append((f"blocks.{i}", _blocks_pred(i)))
        else:
            for tag in ("stage4_blocks", "fuse4", "stage3_blocks", "fuse3", "stage2_blocks", "fuse2", "stage1_blocks"):
                # gradients of everything ordered at or before `tag` are complete when backward passes the sync
                # point placed BEFORE that module group in forward
                stage_tags.append((tag, _prefix_pred(tag)))
        if self.sync_tags is not None:
            stage_tags = [(t, p) for t, p in stage_tags if t in self.sync_tags]
        self.reducer.plan(names, offsets, stage_tags)
        self.reducer.attach(trainer.flat_g)

    def after_step(self):
        self._steps += 1
        if self.bank_sync == "local" and self._steps % self.bank_every == 0 and hasattr(self.model, "global_bank"):
            dist.broadcast(self.model.global_bank.global_k.data, src=0, group=self.group)
            dist.broadcast(self.model.global_bank.global_v.data, src=0, group=self.group)


_ORDER = ("head.", "norm.", "stage4_blocks", "fuse4", "stage3_blocks", "fuse3", "stage2_blocks", "fuse2", "stage1_blocks")


def _prefix_pred(tag: str):
    upto = _ORDER.index(tag)
    ok = _ORDER[: upto + 1]

    def pred(name: str) -> bool:
        return name.startswith(ok)
    return pred


def _blocks_pred(i: int):
    def pred(name: str) -> bool:
        if name.startswith(("head.", "norm.")):
            return True
        if name.startswith("blocks."):
            try:
                return int(name.split(".")[1]) >= i
            except ValueError:
                return False
        return False
    return pred
