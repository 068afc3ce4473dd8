"""The CPU oracle (oracle/qavit_oracle.py) against vectors recorded from the REAL reference
(tests/golden/make_golden.py).  This is what pins the oracle: fp32, tolerance 1e-5 max-rel."""
import numpy as np
import pytest
import torch

from conftest import HQA_TAGS, MODELS, max_rel, sig, zero_by_construction

TOL = 1e-5

# reference module name -> oracle tap key
TAP_MAP_HQA = {
    "pos_drop": "embed", "fuse2": "fuse2", "fuse3": "fuse3", "fuse4": "fuse4",
    "stage1_blocks.0": "stage1_blocks.0", "stage2_blocks.1": "stage2_blocks.1", "stage4_blocks.1": "stage4_blocks.1",
    "stage1_blocks.0.quad_block.swa": "stage1_blocks.0.quad_block.swa",
    "stage1_blocks.0.quad_block.msda": "stage1_blocks.0.quad_block.msda",
    "stage1_blocks.0.quad_block.cga": "stage1_blocks.0.quad_block.cga",
    "stage1_blocks.0.quad_block.cross_attn": "stage1_blocks.0.quad_block.cross_attn",
}
TAP_MAP_Q = {"blocks.0.swa": "blocks.0.swa", "blocks.0.msda": "blocks.0.msda", "blocks.0.cga": "blocks.0.cga",
             "blocks.0.cross_attn": "blocks.0.cross_attn", "blocks.0": "blocks.0", "blocks.7": "blocks.7"}


def _state(Q, tag, **kw):
    build = MODELS[tag][0]
    model = build(Q, **kw)
    Q.fill_module(model)
    return model, {k: v.clone() for k, v in model.state_dict().items()}


@pytest.mark.parametrize("tag", list(MODELS))
def test_eval_forward_matches_reference(tag, golden, Q, oracle):
    torch.set_num_threads(8)
    model, P = _state(Q, tag)
    fwd = getattr(oracle, MODELS[tag][1])
    x = torch.from_numpy(golden[f"{tag}/x"])
    y = torch.from_numpy(golden[f"{tag}/y"])
    taps = {}
    with torch.no_grad():
        logits = fwd(P, x, model.config, train=False, variant=MODELS[tag][2], taps=taps)
    assert max_rel(logits.numpy(), golden[f"{tag}/eval_logits"]) <= TOL
    loss = oracle.loss_fn(logits, y, MODELS[tag][3]).item()
    assert abs(loss - float(golden[f"{tag}/eval_loss"])) <= 1e-5 * abs(float(golden[f"{tag}/eval_loss"]))
    tmap = TAP_MAP_HQA if tag in HQA_TAGS else TAP_MAP_Q
    checked = 0
    for ref_name, key in tmap.items():
        gk = f"{tag}/tap/{ref_name}"
        if gk in golden.files and key in taps:
            assert max_rel(sig(taps[key]), golden[gk]) <= 5e-5, ref_name
            checked += 1
    assert checked >= 5


@pytest.mark.parametrize("tag", list(MODELS))
def test_train_forward_backward_matches_reference(tag, golden, Q, oracle):
    """dropout = drop_path = 0: logits, loss, bank after the in-forward writes, update_count, which
    parameters stay without gradient, and every parameter's gradient norm."""
    torch.set_num_threads(8)
    model, P = _state(Q, tag, dropout=0.0, drop_path=0.0)
    canon = dict(model.named_parameters())
    for k in P:                                    # re-alias: the bank must be ONE tensor under all its keys
        if k.endswith(("global_bank.global_k", "global_bank.global_v", "global_bank.update_count")):
            P[k] = P["global_bank." + k.rsplit(".", 1)[-1]]
    for n in canon:
        P[n].requires_grad_(True)
    fwd = getattr(oracle, MODELS[tag][1])
    x = torch.from_numpy(golden[f"{tag}/x"])
    y = torch.from_numpy(golden[f"{tag}/y"])
    kw = dict(cat_dropout=False, stem_drop=False) if tag in HQA_TAGS else {}
    logits = fwd(P, x, model.config, train=True, variant=MODELS[tag][2], **kw)
    loss = oracle.loss_fn(logits, y, MODELS[tag][3])
    loss.backward()
    assert max_rel(logits.detach().numpy(), golden[f"{tag}/train_logits"]) <= TOL
    assert abs(loss.item() - float(golden[f"{tag}/train_loss"])) <= 1e-5 * abs(float(golden[f"{tag}/train_loss"]))
    assert max_rel(P["global_bank.global_k"].detach().numpy(), golden[f"{tag}/bank_k_after"]) <= TOL
    assert max_rel(P["global_bank.global_v"].detach().numpy(), golden[f"{tag}/bank_v_after"]) <= TOL
    if f"{tag}/update_count" in golden.files:
        assert int(P["global_bank.update_count"]) == int(golden[f"{tag}/update_count"])
    nograd = sorted(n for n in canon if P[n].grad is None)
    assert nograd == sorted(golden[f"{tag}/nograd_names"].tolist())
    names = golden[f"{tag}/grad_names"].tolist()
    norms = np.array([P[n].grad.norm().item() for n in names])
    ref = golden[f"{tag}/grad_norms"]
    # per-parameter relative error (small-gradient tensors are not hidden behind the largest norm)
    err = np.abs(norms - ref) / (ref + 1e-5 * ref.max())   # floor: d/d(upsample bias) is exactly 0 (LN), pure round-off
    # d/d(token_upmix.upsample_attn.bias) is identically 0 (the LayerNorm that follows removes the row mean):
    # the reference's value is pure round-off, so it is not compared
    for i, n in enumerate(names):
        if zero_by_construction(n):
            err[i] = 0.0
    worst = [(float(err[i]), names[i]) for i in np.argsort(-err)[:3]]
    assert err.max() <= 2e-4, worst
    for n in ("head.weight", "pos_embed", "global_bank.global_k", "patch_embed.proj.weight"):
        assert max_rel(P[n].grad.reshape(-1)[:256].numpy(), golden[f"{tag}/grad/{n}"]) <= 5e-5, n
