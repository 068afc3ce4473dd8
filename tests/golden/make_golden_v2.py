#!/usr/bin/env python3
"""Second set of golden vectors (tests/golden/golden_v2.npz), again from the REAL reference on CPU:

  * ``c100v2``  -- HQAViTv2_CIFAR100.HQAViT (ConvNeXt-Tiny style stem, layer-scaled ConvNeXt blocks) at 32x32;
  * ``q224``    -- QAViT.QAViT at its own default size (224 px, patch 16: N=196, 7x7 windows);
  * ``v2_224``  -- QAViTv2.QAViT at the same size (model slice exec'd as text, as in make_golden.py).

Same recording scheme as make_golden.py (eval logits / loss / tap signatures, and a train-mode forward + backward with
every stochastic layer silenced).  The v2 stem hard-wires DropPath(0.1) into four of its blocks
(HQAViTv2_CIFAR100.py:784-785, :797-798): those are silenced too for the train-mode record.
Runs only in the build container (needs /root/reference).  Usage:  python tests/golden/make_golden_v2.py
"""
import io
import os
import sys
from contextlib import redirect_stdout

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.dont_write_bytecode = True
import make_golden as G  # noqa: E402


def _silence_drop_path(model):
    for m in model.modules():
        if type(m).__name__ == "DropPath":
            m.drop_prob = 0.0


def run_model(tag, build, x, y, tap_names, out, ls):
    """make_golden.run_model with DropPath modules silenced in the train-mode pass."""
    filler = G._load_filler()
    with redirect_stdout(io.StringIO()):
        model = build()
    filler.fill_module(model)
    model.eval()
    taps, hooks = {}, []
    mods = dict(model.named_modules())
    for name in tap_names:
        hooks.append(mods[name].register_forward_hook(lambda m, i, o, n=name: taps.__setitem__(n, o)))
    with torch.no_grad():
        logits = model(x)
    for h in hooks:
        h.remove()
    out[f"{tag}/x"] = x.numpy()
    out[f"{tag}/y"] = y.numpy()
    out[f"{tag}/eval_logits"] = logits.numpy()
    out[f"{tag}/eval_loss"] = np.float32(torch.nn.functional.cross_entropy(logits, y, label_smoothing=ls).item())
    for n, t in taps.items():
        out[f"{tag}/tap/{n}"] = G.sig(t[0] if isinstance(t, tuple) else t)
    out[f"{tag}/n_params"] = np.int64(sum(p.numel() for p in model.parameters()))
    out[f"{tag}/state_keys"] = np.array(sorted(model.state_dict().keys()))
    with redirect_stdout(io.StringIO()):
        model = build(dropout=0.0, drop_path=0.0)
    filler.fill_module(model)
    model.train()
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout):
            m.p = 0.0
    _silence_drop_path(model)
    logits = model(x)
    loss = torch.nn.functional.cross_entropy(logits, y, label_smoothing=ls)
    loss.backward()
    out[f"{tag}/train_logits"] = logits.detach().numpy()
    out[f"{tag}/train_loss"] = np.float32(loss.item())
    out[f"{tag}/bank_k_after"] = model.global_bank.global_k.detach().numpy()
    out[f"{tag}/bank_v_after"] = model.global_bank.global_v.detach().numpy()
    if hasattr(model.global_bank, "update_count"):
        out[f"{tag}/update_count"] = np.int64(int(model.global_bank.update_count))
    names, norms, nograd = [], [], []
    for n, p in model.named_parameters():
        if p.grad is None:
            nograd.append(n)
        else:
            names.append(n)
            norms.append(p.grad.norm().item())
    out[f"{tag}/grad_names"] = np.array(names)
    out[f"{tag}/grad_norms"] = np.array(norms, dtype=np.float32)
    out[f"{tag}/nograd_names"] = np.array(nograd)
    for n in ("head.weight", "pos_embed", "global_bank.global_k", "patch_embed.proj.weight"):
        p = dict(model.named_parameters())[n]
        out[f"{tag}/grad/{n}"] = p.grad.reshape(-1)[:256].numpy().copy()


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    c100, tin, q1, v2 = G._import_reference()
    with redirect_stdout(io.StringIO()):
        import HQAViTv2_CIFAR100 as hv2
    g = torch.Generator().manual_seed(4321)
    x32 = torch.randn(4, 3, 32, 32, generator=g)
    y32 = torch.randint(0, 100, (4,), generator=g)
    x224 = torch.randn(2, 3, 224, 224, generator=g)
    y224 = torch.randint(0, 100, (2,), generator=g)
    blk0 = "stage1_blocks.0"
    hqa_taps = ["patch_embed", "cnn_stem.stem", "cnn_stem.stage2", "cnn_stem.downsample2", "cnn_stem.stage3", "cnn_stem.stage4",
                "lmfa2", "rrcv2", "rrcv4", f"{blk0}.quad_block", blk0, "fuse2", "fuse3", "fuse4", "stage4_blocks.1", "norm"]
    q_taps = ["patch_embed", "pos_drop", "blocks.0.swa", "blocks.0.msda", "blocks.0.cga", "blocks.0.cross_attn",
              "blocks.0.ccf_ffn", "blocks.0", "blocks.7", "norm"]
    out = {}
    run_model("c100v2", lambda **kw: hv2.HQAViT(hv2.HQAViTConfig(**kw)), x32, y32, hqa_taps, out, 0.12)
    run_model("q224", lambda **kw: q1.QAViT(q1.QAViTConfig(**kw)), x224, y224, q_taps, out, 0.1)
    run_model("v2_224", lambda **kw: v2.QAViT(v2.QAViTConfig(**kw)), x224, y224, q_taps, out, 0.1)
    path = os.path.join(HERE, "golden_v2.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
