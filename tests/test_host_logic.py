"""CPU-side logic of the product: module tree / state_dict contract, filler, schedules, parameter bookkeeping."""
import math
import os

import numpy as np
import pytest
import torch

from conftest import MODELS


def test_state_dict_layout_matches_reference(golden, Q):
    for tag, (build, *_rest) in MODELS.items():
        m = build(Q)
        keys = sorted(m.state_dict().keys())
        assert keys == golden[f"{tag}/state_keys"].tolist(), tag
        assert sum(p.numel() for p in m.parameters()) == int(golden[f"{tag}/n_params"]), tag
    m = MODELS["c100"][0](Q)
    assert len(m.state_dict()) == 1116                      # SURVEY.md section 8b [probe]
    assert len(list(m.parameters())) == 815
    # the bank is ONE module aliased under every branch (strict load_state_dict needs the aliased keys)
    assert m.stage1_blocks[0].quad_block.swa.global_bank is m.global_bank
    assert m.stage4_blocks[1].quad_block.cross_attn.global_bank.global_k is m.global_bank.global_k


def test_param_group_sizes_known_answers(golden, Q):
    """Name-pattern grouping of HQAViT_C100_Finetune.py:201-221 reproduces the group sizes printed in the reference's
    own fine-tune log ("log hqavit. finetunetxt.txt":19-27), including the 'stage1' substring quirk."""
    m = MODELS["c100"][0](Q)
    groups = {k: 0 for k in ("head", "stage4", "stage3", "stage2", "stage1", "fusion", "cnn_stem", "embeddings", "remaining")}
    for n, p in m.named_parameters():
        if "head" in n:
            groups["head"] += p.numel()
        elif "stage4" in n:
            groups["stage4"] += p.numel()
        elif "stage3" in n:
            groups["stage3"] += p.numel()
        elif "stage2" in n:
            groups["stage2"] += p.numel()
        elif "stage1" in n:
            groups["stage1"] += p.numel()
        elif "fuse" in n or "rrcv" in n or "lmfa" in n:
            groups["fusion"] += p.numel()
        elif "cnn_stem" in n:
            groups["cnn_stem"] += p.numel()
        elif "patch_embed" in n or "pos_embed" in n or "global_bank" in n:
            groups["embeddings"] += p.numel()
        else:
            groups["remaining"] += p.numel()
    assert list(groups.values()) == golden["c100/known_group_sizes"].tolist()
    assert sum(groups.values()) == 6472037


def test_filler_is_deterministic_and_key_seeded(Q):
    a, b = MODELS["c100"][0](Q), MODELS["c100"][0](Q)
    Q.fill_module(a)
    Q.fill_module(b)
    for (n, p), (_, q) in zip(a.state_dict().items(), b.state_dict().items()):
        assert torch.equal(p, q), n
    assert not torch.equal(a.head.weight, a.stage1_blocks[0].quad_block.swa.proj.weight[:100])
    assert int(a.global_bank.update_count) == 0
    assert float(a.cnn_stem.stem[1].running_var.min()) >= 0.5


def test_never_trained_matches_reference_nograd_set(golden, Q):
    harness = __import__("importlib").import_module("qa-vit_amd.harness")
    for tag in ("c100", "tin", "v2_32"):
        m = MODELS[tag][0](Q)
        mine = sorted(n for n, _ in m.named_parameters() if harness.never_trained(n))
        assert mine == sorted(golden[f"{tag}/nograd_names"].tolist()), tag
    assert len([n for n, _ in MODELS["c100"][0](Q).named_parameters() if harness.never_trained(n)]) == 54


def test_onecycle_closed_form_matches_torch(golden, Q):
    harness = __import__("importlib").import_module("qa-vit_amd.harness")
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=6e-4)
    total, warm = 100, 10
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=6e-4, total_steps=total, pct_start=warm / total,
                                                anneal_strategy="cos", div_factor=25.0, final_div_factor=1e4)
    for i in range(total):
        assert math.isclose(harness.onecycle_lr(i, total, 6e-4, warm / total), opt.param_groups[0]["lr"], rel_tol=1e-9, abs_tol=1e-15), i
        opt.step()
        if i < total - 1:
            sched.step()
    assert np.allclose([harness.onecycle_lr(i, 100, 6e-4, 0.1) for i in range(3)], golden["harness/lr"], rtol=1e-9)


def test_config_defaults_match_reference_dataclasses(Q):
    c = Q.HQAViTConfig()
    assert (c.img_size, c.patch_size, c.embed_dim, c.depth, c.num_heads, c.num_learned_tokens, c.linformer_k) == (32, 4, 192, 8, 4, 16, 32)
    assert (c.dropout, c.drop_path, c.window_size, c.dilation_factors, c.num_channel_groups) == (0.1, 0.1, 4, (1, 2), 6)
    t = Q.HQAViTTinyINConfig()
    assert (t.img_size, t.num_classes, t.depth, t.drop_path, t.num_learned_tokens) == (64, 200, 12, 0.2, 64)
    q = Q.QAViTConfig()
    assert (q.img_size, q.patch_size, q.window_size, q.dilation_factors, q.linformer_k) == (224, 16, 7, (1, 2, 3), 64)


def test_model_refuses_cpu_tensors(Q):
    m = MODELS["c100"][0](Q)
    try:
        m(torch.randn(1, 3, 32, 32))
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("a CPU forward must fail loudly")


def test_bucket_plan_covers_flat_buffer_in_backward_order(Q):
    par = __import__("importlib").import_module("qa-vit_amd.parallel")
    m = MODELS["c100"][0](Q)
    named = par.bucket_order(list(m.named_parameters()))
    names = [n for n, _ in named]
    # head first, stage4 before stage1, bank / stem / embeddings last
    assert names[0].startswith(("head.", "norm."))
    assert names.index("stage4_blocks.0.quad_block.norm1.weight") < names.index("stage1_blocks.0.quad_block.norm1.weight")
    assert names.index("stage1_blocks.1.token_upmix.norm.bias") < names.index("global_bank.global_k")
    offs = [0]
    for _, p in named:
        offs.append(offs[-1] + p.numel())

    class FakeGroup:
        pass
    red = par.GradReducer.__new__(par.GradReducer)
    red.bucket_bytes, red.bounds, red.ready_at = 4 << 20, [], {}
    tags = [(t, par._prefix_pred(t)) for t in ("stage4_blocks", "fuse4", "stage3_blocks", "fuse3", "stage2_blocks", "fuse2", "stage1_blocks")]
    bounds = red.plan(names, offs, tags)
    assert bounds[0][0] == 0 and bounds[-1][1] == offs[-1]
    assert all(a[1] == b[0] for a, b in zip(bounds[:-1], bounds[1:]))
    ready = [red.ready_at[t] for t, _ in tags]
    assert ready == sorted(ready) and ready[0] >= 1 and ready[-1] <= len(bounds)
    # a bucket released at tag k only holds parameters whose gradients are complete at tag k
    for (t, pred) in tags:
        upto = bounds[red.ready_at[t] - 1][1] if red.ready_at[t] else 0
        i = 0
        while offs[i + 1] <= upto:
            assert pred(names[i]), (t, names[i])
            i += 1


def test_mix_plan_follows_the_reference_decisions(Q):
    """harness.mix_plan against a scalar restatement of train_epoch's CutMix/MixUp branch and rand_bbox
    (HQAViT_CIFAR100.py:1339-1363, 1378-1399) for the same uniform / beta draws."""
    import math
    import numpy as np
    cfg = Q.TrainingConfig()
    H = W = 32
    rng = np.random.default_rng(5)
    for _ in range(200):
        u = rng.random(4)
        lam_c, lam_m = rng.beta(cfg.cutmix_alpha, cfg.cutmix_alpha), rng.beta(cfg.mixup_alpha, cfg.mixup_alpha)
        # reference control flow with the draws substituted for np.random.*
        mode, lam, box = 0, 1.0, (0, 0, 0, 0)
        if cfg.use_cutmix and u[0] < cfg.mix_prob:
            cut_rat = math.sqrt(1.0 - lam_c)
            cut_w, cut_h = int(W * cut_rat), int(H * cut_rat)
            cx, cy = int(u[2] * W), int(u[3] * H)                      # np.random.randint(W) from a uniform
            x1, y1 = int(np.clip(cx - cut_w // 2, 0, W)), int(np.clip(cy - cut_h // 2, 0, H))
            x2, y2 = int(np.clip(cx + cut_w // 2, 0, W)), int(np.clip(cy + cut_h // 2, 0, H))
            mode, lam, box = 1, 1.0 - ((x2 - x1) * (y2 - y1) / float(W * H)), (x1, y1, x2, y2)
        elif cfg.use_mixup and u[1] < cfg.mix_prob:
            mode, lam = 2, lam_m
        plan = Q.mix_plan(torch.tensor(u, dtype=torch.float64), torch.tensor(lam_c, dtype=torch.float64), torch.tensor(lam_m, dtype=torch.float64), cfg, H, W)
        assert int(plan[0]) == mode
        assert abs(float(plan[1]) - lam) <= 1e-6
        if mode == 1:
            assert tuple(int(v) for v in plan[2:]) == box


def test_reference_named_shims_resolve(golden, Q):
    """qa-vit_amd/shims/<reference module name>.py: the classes the reference's scripts import by name, with the
    reference's one-argument constructors and state_dict layouts."""
    import importlib.util
    import os
    sdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qa-vit_amd", "shims")
    want = {"HQAViT_CIFAR100": ("HQAViT", "HQAViTConfig", "c100"), "HQAViT_IN_Tiny": ("HQAViT", "HQAViTConfig", "tin"),
            "HQAViTv2_CIFAR100": ("HQAViT", "HQAViTConfig", "c100v2"), "QAViT": ("QAViT", "QAViTConfig", "q224"),
            "QAViTv2": ("QAViT", "QAViTConfig", "v2_224"), "QAViTV2_EXTREME": ("QAViT", "QAViTConfig", None)}
    for name, (cls, cfg, tag) in want.items():
        spec = importlib.util.spec_from_file_location("shim_" + name, os.path.join(sdir, name + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        model = getattr(mod, cls)(getattr(mod, cfg)())
        if tag is not None:
            assert sorted(model.state_dict().keys()) == golden[f"{tag}/state_keys"].tolist(), name


# ---------------------------------------------------------------------------------------------------
# round 2: config 1's per-epoch two-group schedule, EMA decay warm-up, init distributions (golden_r2.npz)
# ---------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def golden_r2():
    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_r2.npz"))


def test_finetune_schedule_matches_torch_schedulers(golden_r2, Q):
    """harness.finetune_lr (closed form) == the LR both parameter groups had under the reference's own construction
    (CosineAnnealingLR then LinearLR on one AdamW, stepped per epoch; HQAViT_Tiny_Cifar10.py:481-496, :520-523), recorded
    from torch's schedulers with the script's FineTuneConfig defaults, and the 3-epoch variant of the recipe trace."""
    hp = golden_r2["sched/hparams"]
    cfg = Q.FineTuneConfig()
    assert (cfg.base_lr, cfg.head_lr_multiplier, cfg.min_lr, cfg.epochs, cfg.warmup_epochs) == tuple(hp)
    ref = golden_r2["sched/lr"]
    got = np.array([Q.harness.finetune_lr(e, cfg) for e in range(ref.shape[0])])
    assert np.abs(got - ref).max() <= 1e-12 + 1e-9 * ref.max(), (got, ref)
    hp1 = golden_r2["cfg1/hparams"]
    cfg1 = Q.FineTuneConfig(epochs=int(hp1[6]), warmup_epochs=int(hp1[7]))
    ref1 = golden_r2["cfg1/lr"]
    got1 = np.array([Q.harness.finetune_lr(e, cfg1) for e in range(ref1.shape[0])])
    assert np.abs(got1 - ref1).max() <= 1e-12 + 1e-9 * ref1.max(), (got1, ref1)


def test_ema_decay_warmup_rule(Q):
    """HQAViT_CIFAR100.py:1634-1638: decay = 0.99 + (0.999 - 0.99) * epoch / warmup_epochs for epoch <= warmup_epochs."""
    cfg = Q.TrainingConfig()
    assert cfg.use_ema and cfg.ema_decay == 0.999 and cfg.ema_decay_warmup == 0.99 and cfg.warmup_epochs == 20
    f = Q.harness.ema_decay_for_epoch
    assert abs(f(1, cfg) - (0.99 + 0.009 / 20)) < 1e-12 and abs(f(10, cfg) - 0.9945) < 1e-12
    assert f(20, cfg) == pytest.approx(0.999) and f(21, cfg) == 0.999 and f(400, cfg) == 0.999


@pytest.mark.parametrize("tag", ["c100", "q32"])
def test_init_distributions_match_reference(tag, golden_r2, Q):
    """HQAViT.__init__/_init_weights (HQAViT_CIFAR100.py:1146-1224; `self.apply` runs last, so every Conv2d -- the custom
    depthwise init of :664-666 included -- ends up kaiming_normal(fan_out), every Linear trunc_normal(0.02) with zero bias,
    every LayerNorm (1, 0)): per-parameter moments of a fresh model here against the moments recorded from a fresh
    REFERENCE model.  Constants must be equal; random tensors must agree in mean and std within sampling error."""
    torch.manual_seed(1)
    if tag == "c100":
        model = Q.HQAViT(Q.HQAViTConfig())
    else:
        model = Q.QAViT(Q.qavit32_config(), "v1")
    names = golden_r2[f"init/{tag}/names"].tolist()
    mom = golden_r2[f"init/{tag}/moments"]
    params = dict(model.named_parameters())
    assert list(params) == names
    bad = []
    for n, (mean, std, lo, hi, numel) in zip(names, mom):
        t = params[n].detach().float()
        assert t.numel() == int(numel), n
        m2 = float(t.mean())
        s2 = float(t.std()) if t.numel() > 1 else 0.0
        if std == 0.0 or lo == hi:                                   # a constant fill
            if not (float(t.min()) == lo and float(t.max()) == hi):
                bad.append((n, "const", lo, hi, float(t.min()), float(t.max())))
            continue
        k = float(numel)
        tol_mean = 6.0 * std * np.sqrt(2.0 / k) + 1e-12              # both sides are samples
        tol_std = 6.0 / np.sqrt(2.0 * k) * np.sqrt(2.0) + 0.0
        if abs(m2 - mean) > tol_mean or abs(s2 / std - 1.0) > max(tol_std, 0.02 if k >= 4096 else tol_std):
            bad.append((n, "moments", mean, std, m2, s2, k))
        if not (float(t.abs().max()) <= 1.6 * max(abs(lo), abs(hi)) + 1e-12 or k < 64):
            bad.append((n, "range", lo, hi, float(t.min()), float(t.max())))
    assert not bad, bad[:8]


def test_no_stock_op_fallbacks_on_the_hot_path(Q):
    """Shapes the HIP kernels do not cover raise instead of silently taking a stock torch op: a BatchNorm bnorm.hip does not cover
    (no affine parameters / no running statistics / a channel count its vectors do not tile), LMFAdapter's bilinear resize
    (HQAViT_CIFAR100.py:840-842, reached by no shipped configuration), SWA window padding (:424-428)."""
    import importlib
    import pytest
    M = importlib.import_module("qa-vit_amd.modules")
    t = torch.zeros(2, 64, 36)
    for bn in (torch.nn.BatchNorm2d(36, affine=False), torch.nn.BatchNorm2d(36, track_running_stats=False), torch.nn.BatchNorm2d(36, momentum=None)):
        with pytest.raises(NotImplementedError, match="no HIP kernel"):
            M._bn_tokens(t, bn, True)
    with pytest.raises(NotImplementedError, match="no HIP kernel"):
        M._bn_tokens(torch.zeros(2, 64, 30), torch.nn.BatchNorm2d(30), True)          # 30 channels: not a multiple of the fp32 vector
    cfg = Q.HQAViTConfig()
    bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim)
    swa = M.EfficientSpatialWindowAttention(cfg, bank, M._Ctx("hqa"))
    with pytest.raises(NotImplementedError, match="window padding"):
        swa(torch.zeros(1, 36, cfg.embed_dim))                                        # 6x6 grid, window 4
    src = open(M.__file__).read()
    assert "TF." not in src and "torch.nn.functional" not in src                     # the module file holds no stock functional op at all
