"""Lean training harness: the counterpart of the reference's ``train_epoch`` (HQAViT_CIFAR100.py:1366-1458)
and optimiser set-up (:1566-1583) without its host-sync storm.

What is reproduced: AdamW(betas .9/.999, wd 0.06) over ``model.parameters()`` --
tensors that never receive a gradient are skipped exactly as torch skips ``grad is None``; OneCycleLR
(max_lr, pct_start = warmup/total, cos, div 25, final_div 1e4) stepped per iteration; CrossEntropy with label
smoothing; per-tensor clip 0.1 for names containing ``cnn_stem`` / ``dwconv`` then global clip 0.5; EMA
``p_ema = d*p_ema + (1-d)*p`` with buffers copied.

MI355X-first structure: parameters, gradients and Adam moments live in FLAT fp32 buffers (parameters and
``.grad`` are views), so clipping + AdamW is two kernels, gradient all-reduce works on contiguous buckets, and
the whole step (weight re-pack, forward, backward, optimiser) is capturable in one hipGraph.

Two recipes: ``TrainingConfig`` = HQAViT_CIFAR100.py's pre-training loop (OneCycle per iteration, two clips, EMA);
``FineTuneConfig`` = HQAViT_Tiny_Cifar10.py's transfer loop (BASELINE config 1): two parameter groups (names containing
``head`` at base_lr * head_lr_multiplier, :327-342), LinearLR(0.1 -> 1) warm-up then CosineAnnealingLR stepped once per
EPOCH (:384, :481-496, :520-523 -- call ``Trainer.epoch_end()``), one global clip at 1.0, label smoothing 0.1, no EMA.
"""
import os
import math
import time
from copy import deepcopy
from dataclasses import dataclass
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as TF

from . import functional as F
from . import kernels as K


@dataclass
class TrainingConfig:
    """Subset of HQAViT_CIFAR100.TrainingConfig (:81-122) that the step consumes, same names/defaults."""
    batch_size: int = 256
    epochs: int = 450
    warmup_epochs: int = 20
    base_lr: float = 6e-4
    weight_decay: float = 0.06
    label_smoothing: float = 0.12
    max_grad_norm: float = 0.5
    local_clip: float = 0.1          # per-tensor clip for 'cnn_stem' / 'dwconv' (:1416-1418)
    use_amp: bool = True
    amp_dtype: str = "bfloat16"
    use_ema: bool = True
    ema_decay: float = 0.999
    ema_decay_warmup: float = 0.99   # decay ramps from this to ema_decay over the warm-up epochs (:1634-1638)
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    # batch mixing (:118-122); ``device_mix`` turns the device-side implementation on in Trainer
    use_mixup: bool = True
    mixup_alpha: float = 0.9
    use_cutmix: bool = True
    cutmix_alpha: float = 1.0
    mix_prob: float = 0.6
    device_mix: bool = False


@dataclass
class FineTuneConfig:
    """HQAViT_Tiny_Cifar10.FineTuneConfig (:34-64), the fields the step consumes, same names/defaults."""
    batch_size: int = 256
    epochs: int = 100
    warmup_epochs: int = 5
    base_lr: float = 1e-4
    head_lr_multiplier: float = 10.0
    min_lr: float = 1e-6
    weight_decay: float = 0.05
    label_smoothing: float = 0.1
    max_grad_norm: float = 1.0
    use_amp: bool = True
    amp_dtype: str = "bfloat16"
    beta1: float = 0.9
    beta2: float = 0.999
    eps: float = 1e-8
    # absent from the transfer recipe (kept so Trainer reads one interface)
    local_clip: float = 0.0
    use_ema: bool = False
    device_mix: bool = False


def finetune_lr(epoch_idx: int, cfg: "FineTuneConfig"):
    """LR of the (backbone, head) groups during 0-based epoch ``epoch_idx`` = after ``epoch_idx`` per-epoch scheduler
    steps of HQAViT_Tiny_Cifar10.main's pair of schedulers (:481-496, :520-523): LinearLR(start 0.1, end 1,
    total_iters = warmup_epochs) for the first warmup_epochs steps, then CosineAnnealingLR(T_max = epochs - warmup_epochs,
    eta_min = min_lr -- the same ABSOLUTE floor for both groups) in its chainable form, which from lr = base equals the
    closed form below."""
    warm, tmax = cfg.warmup_epochs, cfg.epochs - cfg.warmup_epochs
    out = []
    for base in (cfg.base_lr, cfg.base_lr * cfg.head_lr_multiplier):
        if epoch_idx <= warm:
            f = 0.1 + 0.9 * (epoch_idx / warm) if warm > 0 else 1.0
            out.append(base * f)
        else:
            t = epoch_idx - warm
            out.append(cfg.min_lr + (base - cfg.min_lr) * (1.0 + math.cos(math.pi * t / tmax)) / 2.0)
    return tuple(out)


def ema_decay_for_epoch(epoch: int, cfg: "TrainingConfig") -> float:
    """EMA decay during 1-based ``epoch`` (HQAViT_CIFAR100.py:1634-1638): linear ramp over the warm-up epochs."""
    if epoch <= cfg.warmup_epochs:
        return cfg.ema_decay_warmup + (cfg.ema_decay - cfg.ema_decay_warmup) * (epoch / cfg.warmup_epochs)
    return cfg.ema_decay


def mix_plan(u: torch.Tensor, lam_cut: torch.Tensor, lam_mix: torch.Tensor, cfg: "TrainingConfig", H: int, W: int) -> torch.Tensor:
    """The per-step decisions of train_epoch's CutMix / MixUp branch (HQAViT_CIFAR100.py:1378-1399, rand_bbox :1339-1363)
    as tensor arithmetic, so they can live on the device and inside a captured step.  ``u`` = 4 uniforms in [0,1)
    (cutmix coin, mixup coin, box centre x, box centre y); ``lam_cut`` / ``lam_mix`` = the Beta(alpha, alpha) draws.
    -> float[6]: mode (0 none / 1 cutmix / 2 mixup), lambda, x1, y1, x2, y2."""
    cut = (u[0] < cfg.mix_prob) if cfg.use_cutmix else torch.zeros((), dtype=torch.bool, device=u.device)
    mixu = (~cut) & ((u[1] < cfg.mix_prob) if cfg.use_mixup else torch.zeros((), dtype=torch.bool, device=u.device))
    cut_rat = torch.sqrt(1.0 - lam_cut)
    cut_w = torch.floor(W * cut_rat)
    cut_h = torch.floor(H * cut_rat)
    cx = torch.floor(u[2] * W)
    cy = torch.floor(u[3] * H)
    hw, hh = torch.floor(cut_w / 2), torch.floor(cut_h / 2)
    x1, x2 = torch.clamp(cx - hw, 0, W), torch.clamp(cx + hw, 0, W)
    y1, y2 = torch.clamp(cy - hh, 0, H), torch.clamp(cy + hh, 0, H)
    lam_box = 1.0 - (x2 - x1) * (y2 - y1) / float(W * H)           # "adjust lambda to exactly match pixel ratio" (:1390)
    one = torch.ones((), dtype=torch.float32, device=u.device)
    lam = torch.where(cut, lam_box, torch.where(mixu, lam_mix, one))
    mode = cut.float() + 2.0 * mixu.float()
    return torch.stack([mode, lam, x1, y1, x2, y2]).float()


def never_trained(name: str) -> bool:
    """Parameters the reference's forward never puts on the autograd path: every ``global_bank.write_*`` (the
    write runs on ``.data`` under no consumers) and the branch ``norm`` that only feeds the bank write
    (SURVEY.md section 7 'Unused parameters': 54 tensors on HQA-ViT C100)."""
    if "global_bank.write_" in name:
        return True
    for br in (".swa.norm.", ".msda.norm.", ".cga.norm."):
        if br in name:
            return True
    return False


def onecycle_lr(step: int, total: int, max_lr: float, pct_start: float, div: float = 25.0, final_div: float = 1e4) -> float:
    """torch.optim.lr_scheduler.OneCycleLR (cos, two phases) in closed form; ``step`` = number of scheduler steps taken."""
    initial, minimum = max_lr / div, max_lr / div / final_div
    up_end = float(pct_start * total) - 1.0
    down_end = float(total) - 1.0

    def cos(a, b, pct):
        return b + (a - b) / 2.0 * (math.cos(math.pi * pct) + 1.0)
    if step <= up_end:
        return cos(initial, max_lr, step / up_end if up_end > 0 else 1.0)
    return cos(max_lr, minimum, (step - up_end) / (down_end - up_end))


class ModelEMA:
    """HQAViT_CIFAR100.py:128-184."""

    def __init__(self, model: nn.Module, decay: float = 0.9999, device: Optional[str] = None):
        self.ema = deepcopy(model).eval()
        for p in self.ema.parameters():
            p.requires_grad_(False)
        self.decay = decay
        if device is not None:
            self.ema.to(device)

    @torch.no_grad()
    def update(self, model: nn.Module):
        ep = [p for _, p in self.ema.named_parameters()]
        mp = [p.detach() for _, p in model.named_parameters()]
        torch._foreach_mul_(ep, self.decay)
        torch._foreach_add_(ep, mp, alpha=1.0 - self.decay)
        mb = dict(model.named_buffers())
        for n, b in self.ema.named_buffers():
            if n in mb:
                b.copy_(mb[n])

    def set_decay(self, decay: float):
        self.decay = decay


class GradientMonitor:
    """HQAViT_CIFAR100.py:190-250 surface (``log_gradients`` / ``check_explosion``) without the ~3 k host syncs per
    call: one fused norm over all gradients and one over all parameters (2 syncs); per-layer statistics only when
    ``detailed`` is requested."""

    def __init__(self):
        self.grad_norms, self.param_norms, self.layer_grad_history, self.explosion_count = [], [], {}, 0

    @torch.no_grad()
    def log_gradients(self, model, detailed=False):
        named = [(n, p) for n, p in model.named_parameters() if p.grad is not None]
        if not named:
            return 0.0, 0.0, {}, {}
        gn = torch.stack(torch._foreach_norm([p.grad for _, p in named]))
        pn = torch.stack(torch._foreach_norm([p.detach() for _, p in named]))
        total, param = float(gn.norm()), float(pn.norm())
        self.grad_norms.append(total)
        self.param_norms.append(param)
        layer_stats = {}
        if detailed:
            g, q = gn.tolist(), pn.tolist()
            for (n, _), a, b in zip(named, g, q):
                k = ".".join(n.split(".")[:2])
                st = layer_stats.setdefault(k, {"grad_norm": 0.0, "param_norm": 0.0, "count": 0})
                st["grad_norm"] += a
                st["param_norm"] += b
                st["count"] += 1
            for k, st in layer_stats.items():
                self.layer_grad_history.setdefault(k, []).append(st["grad_norm"] / max(st["count"], 1))
        return total, param, {}, layer_stats

    def check_explosion(self, threshold=50.0):
        bad = bool(self.grad_norms) and self.grad_norms[-1] > threshold
        self.explosion_count += int(bad)
        return bad


class BatchStager:
    """Host -> device batch staging with pinned double buffers on a copy stream: while the step
    consumes slot i, the loader thread fills slot i^1's pinned buffers and the copy engine moves them; the compute
    stream only waits on the slot's event.  ``put`` returns the slot to pass to ``get``.

        st = BatchStager((B, 3, 32, 32), B, device)
        slot = st.put(x_cpu, y_cpu)                 # before the previous step has finished
        x, y = st.get(slot); trainer.replay(x, y)
    """

    def __init__(self, x_shape, n_labels: int, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.hx = [torch.empty(x_shape, dtype=torch.float32).pin_memory() for _ in range(2)]
        self.hy = [torch.empty(n_labels, dtype=torch.int64).pin_memory() for _ in range(2)]
        self.dx = [torch.empty(x_shape, dtype=torch.float32, device=self.device) for _ in range(2)]
        self.dy = [torch.empty(n_labels, dtype=torch.int64, device=self.device) for _ in range(2)]
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self._i = 0
        self._used = [False, False]

    def put(self, x_cpu: torch.Tensor, y_cpu: torch.Tensor) -> int:
        i = self._i
        self._i ^= 1
        if self._used[i]:
            self.consumed[i].synchronize()                 # the step that read this slot's device buffers is done
        self.hx[i].copy_(x_cpu)
        self.hy[i].copy_(y_cpu)
        with torch.cuda.stream(self.stream):
            self.dx[i].copy_(self.hx[i], non_blocking=True)
            self.dy[i].copy_(self.hy[i], non_blocking=True)
            self.ready[i].record(self.stream)
        return i

    def get(self, slot: int):
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.ready[slot])
        self._used[slot] = True
        return self.dx[slot], self.dy[slot]

    def done(self, slot: int):
        """Call after enqueuing the step that consumed ``slot`` (records when its device buffers may be overwritten)."""
        self.consumed[slot].record(torch.cuda.current_stream(self.device))


class Trainer:
    """One object = model + flat optimiser state + (optional) data-parallel reducer + (optional) hipGraph."""

    def __init__(self, model: nn.Module, cfg: TrainingConfig, total_steps: int, warmup_steps: Optional[int] = None,
                 reducer=None, compute_dtype: Optional[torch.dtype] = None, order=None):
        self.model, self.cfg, self.reducer = model, cfg, reducer
        dev = next(model.parameters()).device
        self.device = dev
        if compute_dtype is None:
            compute_dtype = torch.bfloat16 if (cfg.use_amp and cfg.amp_dtype == "bfloat16") else torch.float32
        if hasattr(model, "compute_dtype"):
            model.compute_dtype = compute_dtype
        named = list(model.named_parameters())
        if order is not None:                           # bucket-friendly ordering (parallel.bucket_order)
            named = order(named)
        self.names = [n for n, _ in named]
        self.params = [p for _, p in named]
        sizes = [p.numel() for p in self.params]
        # every tensor starts on a 256-byte boundary of the flat buffers: the kernels' 16-byte vector loads of weights,
        # bank rows and Linformer matrices need aligned bases (padding elements stay 0 through AdamW / all-reduce)
        ALIGN = 64
        self.offsets = [0]
        for s in sizes:
            self.offsets.append((self.offsets[-1] + s + ALIGN - 1) // ALIGN * ALIGN)
        n = self.offsets[-1]
        self.flat_p = torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros(n, dtype=torch.float32, device=dev)
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        skip = torch.ones(n, dtype=torch.uint8, device=dev)          # padding between tensors is never touched
        with torch.no_grad():
            for p, name, o, s in zip(self.params, self.names, self.offsets, sizes):
                self.flat_p[o:o + s].copy_(p.reshape(-1))
                p.data = self.flat_p[o:o + s].view_as(p)
                p.grad = self.flat_g[o:o + s].view_as(p)
                if not (never_trained(name) or not p.requires_grad):
                    skip[o:o + s] = 0
        self.skip = skip
        self.local_clip_params = [p for p, nme in zip(self.params, self.names) if ("cnn_stem" in nme or "dwconv" in nme)]
        segs = [(o, s) for nme, o, s in zip(self.names, self.offsets, sizes) if ("cnn_stem" in nme or "dwconv" in nme)]
        self.local_clip_seg = torch.tensor(segs, dtype=torch.int64, device=dev).reshape(-1, 2) if segs else None
        self.local_clip_ws = torch.zeros(2 * max(len(segs), 1), dtype=torch.float32, device=dev)
        self.total_steps = total_steps
        self.finetune = isinstance(cfg, FineTuneConfig)
        if self.finetune:
            # two parameter groups: names containing 'head' (HQAViT_Tiny_Cifar10.py:333) -> [lo, hi) element ranges of the
            # flat buffers, one AdamW launch per contiguous range; LR table indexed by EPOCH: [epochs + 1, 2]
            self.group_ranges = []                      # (group id 0 backbone / 1 head, lo, hi)
            for nme, o, sz in zip(self.names, self.offsets, sizes):
                gid = 1 if "head" in nme else 0
                if self.group_ranges and self.group_ranges[-1][0] == gid:
                    self.group_ranges[-1] = (gid, self.group_ranges[-1][1], o + sz)
                else:
                    self.group_ranges.append((gid, o, o + sz))
            table = [finetune_lr(e, cfg) for e in range(cfg.epochs + 1)]
            self.lr_table = torch.tensor(table, dtype=torch.float32, device=dev)          # [epochs + 1, 2]
            self.lr_dev = torch.zeros(2, dtype=torch.float32, device=dev)
        else:
            warm = warmup_steps if warmup_steps is not None else max(1, int(total_steps * cfg.warmup_epochs / max(cfg.epochs, 1)))
            table = [onecycle_lr(i, total_steps, cfg.base_lr, warm / total_steps) for i in range(total_steps)]
            self.lr_table = torch.tensor(table, dtype=torch.float32, device=dev)
            self.lr_dev = torch.zeros(1, dtype=torch.float32, device=dev)
        self.step_idx = torch.zeros(1, dtype=torch.int64, device=dev)       # scheduler steps taken (finetune: epochs finished)
        self.step_f = torch.zeros(1, dtype=torch.float32, device=dev)       # Adam step count (1-based at use)
        self.gnorm = torch.zeros(2, dtype=torch.float32, device=dev)        # [this step's global grad norm, running max (NaN sticks)]
        self.partial = torch.zeros(1024, dtype=torch.float32, device=dev)
        self.loss = torch.zeros((), dtype=torch.float32, device=dev)
        self.ema_flat = None
        if cfg.use_ema:
            self.ema_flat = self.flat_p.clone()
            self.ema_w = torch.full((1,), 1.0 - cfg.ema_decay, dtype=torch.float32, device=dev)   # 1 - decay, device-resident: a captured step follows set_ema_decay
        self.graph = None
        self._static_x = self._static_y = None
        self.rt = K.Runtime.get(dev) if dev.type == "cuda" else None
        self._mix_site = K.new_site()

    # ------------------------------------------------------------------------------------------
    def _fwd_bwd(self, x, y):
        K.Stamps.mark("step.begin")
        self.rt.advance()
        F.pack_for(self.device).refresh()
        self.flat_g.zero_()
        if self.reducer is not None:
            self.reducer.begin_step()
        if self.cfg.device_mix:
            x, y_b, lam = self._mix(x, y)
            loss = F.cross_entropy(self.model(x), y, self.cfg.label_smoothing, y_b=y_b, lam=lam.reshape(1).float())
        else:
            loss = F.cross_entropy(self.model(x), y, self.cfg.label_smoothing)
        K.Stamps.mark("bwd.begin")
        loss.backward()
        K.Stamps.mark("bwd.end")
        F.SideStream.join(self.device)                     # weight-gradient GEMMs ran on the side stream
        if self.reducer is not None:
            self.reducer.finish(self.flat_g)
        return loss.detach()

    def _mix(self, x, y):
        """Device-side CutMix / MixUp (no host RNG, no host sync: capturable).  -> mixed x, permuted labels, lambda."""
        cfg = self.cfg
        dev = x.device
        B, _, H, W = x.shape
        u = torch.rand(4, device=dev)

        def beta(a):
            g = torch._standard_gamma(torch.full((2,), float(a), device=dev))
            return g[0] / (g[0] + g[1])
        plan = mix_plan(u, beta(cfg.cutmix_alpha), beta(cfg.mixup_alpha), cfg, H, W)
        perm = torch.empty(B, dtype=torch.int64, device=dev)
        K.rand_perm(perm, B, self.rt.rng, self._mix_site)          # own kernel: a library sort brings memset nodes into the graph
        out = torch.empty_like(x)
        K.mix_apply(x.contiguous().float(), perm, plan, out)
        return out, y[perm], plan[1]

    def _optim(self):
        cfg = self.cfg
        if cfg.local_clip > 0 and self.local_clip_seg is not None:
            K.local_clip(self.flat_g, self.local_clip_seg, cfg.local_clip, self.local_clip_ws)
        K.l2norm(self.flat_g, self.partial, self.gnorm)
        self.step_f += 1.0
        if self.finetune:
            self.lr_dev.copy_(torch.index_select(self.lr_table, 0, torch.clamp(self.step_idx, max=self.lr_table.shape[0] - 1)).reshape(2))
            for gid, lo, hi in self.group_ranges:
                K.adamw(self.flat_p[lo:hi], self.flat_g[lo:hi], self.m[lo:hi], self.v[lo:hi], self.skip[lo:hi], self.lr_dev[gid:gid + 1],
                        cfg.beta1, cfg.beta2, cfg.eps, cfg.weight_decay, self.step_f, self.gnorm, cfg.max_grad_norm)
        else:
            torch.index_select(self.lr_table, 0, torch.clamp(self.step_idx, max=self.total_steps - 1), out=self.lr_dev)
            K.adamw(self.flat_p, self.flat_g, self.m, self.v, self.skip, self.lr_dev, cfg.beta1, cfg.beta2, cfg.eps,
                    cfg.weight_decay, self.step_f, self.gnorm, cfg.max_grad_norm)
            self.step_idx += 1                          # OneCycle: one scheduler step per iteration
        F.pack_for(self.device).mark_stale()            # the packed compute-dtype weights no longer match the parameters
        if self.ema_flat is not None:
            torch.lerp(self.ema_flat, self.flat_p, self.ema_w, out=self.ema_flat)
        K.Stamps.mark("step.end")

    def epoch_end(self):
        """The per-epoch ``scheduler.step()`` of the transfer recipe (HQAViT_Tiny_Cifar10.py:384); no-op for OneCycle."""
        if self.finetune:
            self.step_idx += 1

    def set_ema_decay(self, decay: float):
        """ModelEMA.set_decay (HQAViT_CIFAR100.py:182-184); device-resident, so a captured step follows it."""
        self.ema_w.fill_(1.0 - float(decay))

    @torch.no_grad()
    def ema_model(self) -> nn.Module:
        """The EMA weights as an eval-mode copy of the model (ModelEMA.ema, HQAViT_CIFAR100.py:131-137): parameters from
        the flat EMA buffer, buffers copied from the live model (:151-156)."""
        if self.ema_flat is None:
            raise RuntimeError("cfg.use_ema is False")
        ema = deepcopy(self.model).eval()               # Parameter.__deepcopy__ clones each view: the flat buffers stay behind
        for p in ema.parameters():
            p.grad = None
        ep = dict(ema.named_parameters())
        sizes = [p.numel() for p in self.params]
        for nme, o, sz in zip(self.names, self.offsets, sizes):
            ep[nme].requires_grad_(False)
            ep[nme].copy_(self.ema_flat[o:o + sz].view_as(ep[nme]))
        return ema

    def fwd_bwd(self, x, y):
        """forward + loss + backward (+ gradient all-reduce): the unit BASELINE.json's metric times."""
        self.model.train()
        return self._fwd_bwd(x, y)

    def step(self, x, y):
        """Full optimiser step, eager.  Returns the (device) loss tensor."""
        self.model.train()
        loss = self._fwd_bwd(x, y)
        self._optim()
        return loss

    # ------------------------------------------------------------------------------------------
    def _state(self):
        """Everything a step mutates: flat parameters (the bank included) and Adam moments, step counters, EMA, the model's
        buffers (BatchNorm running statistics, ``update_count``), the dropout RNG words."""
        st = [self.flat_p, self.m, self.v, self.step_f, self.step_idx, self.gnorm, self.rt.rng]
        if self.ema_flat is not None:
            st.append(self.ema_flat)
        st.extend(b for _, b in self.model.named_buffers())
        return st

    def capture(self, x, y, with_optim: bool = True, warmup: int = 3, inspect: bool = False, debug_dot: Optional[str] = None):
        """Capture the whole step into one hipGraph (torch.cuda.CUDAGraph).  ``x``/``y`` give the static shapes.
        The ``warmup`` eager steps that precede the capture (allocator / workspace / table warm-up) run real optimiser
        steps; every piece of training state is snapshotted before and restored after them, so capturing does not
        advance the run (parameters, Adam moments, LR position, bank, BatchNorm statistics, RNG all stay where they were).
        ``inspect``: keep the hipGraph_t and leave its node census (graphinfo.node_kinds) in ``self.graph_nodes``;
        ``debug_dot``: with ``inspect``, path for hipGraphDebugDotPrint's dump of the captured graph."""
        inspect = inspect or bool(debug_dot)
        self.model.train()
        self._static_x, self._static_y = x.clone(), y.clone()
        saved = [t.clone() for t in self._state()]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                loss = self._fwd_bwd(self._static_x, self._static_y)
                if with_optim:
                    self._optim()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if self.reducer is not None and torch.distributed.is_available() and torch.distributed.is_initialized():
            # The process group's watchdog thread still holds the warm-up steps' collectives and polls their end events
            # (hipEventQuery) until it has seen each one complete.  Those events were recorded on the reducer's side stream; once
            # that stream joins the capture, HIP answers such a query with hipErrorCapturedEvent and the watchdog takes the process
            # down.  Everything is complete on the device after the synchronize above: wait until the watchdog has RETIRED its list
            # (read from the flight recorder; raises on a timeout) before the capture begins -- collectives issued DURING capture are
            # never put on that list.
            from . import parallel
            self.watchdog_drain = parallel.drain_watchdog()
        with torch.no_grad():
            for t, v in zip(self._state(), saved):
                t.copy_(v)
        F.pack_for(self.device).mark_stale()
        g = torch.cuda.CUDAGraph(keep_graph=True) if inspect else torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            loss = self._fwd_bwd(self._static_x, self._static_y)
            if with_optim:
                self._optim()
            self.loss.copy_(loss)
        self.graph_nodes = None
        if inspect:
            from . import graphinfo
            raw = g.raw_cuda_graph()
            self.graph_nodes = graphinfo.node_kinds(raw)
            if debug_dot:
                graphinfo.dot_print(raw, debug_dot)
        self.graph = g
        return g

    def replay(self, x=None, y=None):
        if x is not None:
            self._static_x.copy_(x, non_blocking=True)
            self._static_y.copy_(y, non_blocking=True)
        self.graph.replay()
        F.pack_for(self.device).mark_stale()            # the replayed AdamW moved the parameters past the packed copies
        return self.loss

    def grad_norm(self) -> float:
        return float(self.gnorm[0].item())

    def grad_norm_max(self) -> float:
        """Largest global gradient norm any step has seen since construction (NaN / inf stick): lets a test check EVERY
        step of a back-to-back replay run with one host read."""
        return float(self.gnorm[1].item())
