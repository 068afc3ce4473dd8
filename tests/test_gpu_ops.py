"""Operator-level parity of the HIP kernels (through the C-ABI) against plain PyTorch fp32 math on the same
inputs.  fp32 path: <= 2e-5 max-rel forward, <= 2e-4 for gradients reduced with atomics; bf16 path: loose
(bf16 has 8 mantissa bits) -- the parity gate of BASELINE.json is on fp32."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu

DEV = "cuda"


def rel(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-20))


def tol(dtype, fwd=True):
    if dtype == torch.float32:
        return 3e-5 if fwd else 3e-4
    return 3e-2 if fwd else 6e-2


def leaf(*shape, scale=1.0, seed=0):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(DEV).requires_grad_(True)


@pytest.fixture(scope="module")
def F(Q):
    Q.lib.load()
    import importlib
    return importlib.import_module("qa-vit_amd.functional")


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,K,N", [(1000, 192, 576), (4096, 48, 192), (777, 96, 100), (64, 192, 16), (300, 384, 192), (130, 1024, 256), (50, 100, 192)])
def test_linear_plain(F, dtype, M, K, N):
    x = leaf(M, K, seed=1)
    w = leaf(N, K, scale=0.05, seed=2)
    b = leaf(N, scale=0.1, seed=3)
    xr, wr, br = [t.detach().clone().requires_grad_(True) for t in (x, w, b)]
    y = F.linear(x.to(dtype) if dtype != torch.float32 else x, w, b)
    ref = TF.linear(xr, wr, br)
    assert y.dtype == dtype
    assert rel(y, ref) <= tol(dtype)
    g = torch.randn_like(ref)
    y.backward(g.to(dtype))
    ref.backward(g)
    assert rel(x.grad, xr.grad) <= tol(dtype, False)
    assert rel(w.grad, wr.grad) <= tol(dtype, False)
    assert rel(b.grad, br.grad) <= tol(dtype, False)


@pytest.mark.parametrize("mode", ["one_launch", "class_launches"])
def test_weight_gradient_gemm_tile_classes(F, Q, mode):
    """The grouped weight-gradient GEMM on one list holding every tile class (32 .. 256 columns on either side, multi-tile N and K,
    LayerNorm-on-load, ragged M, 74 problems = three writer launches of the device-side table) against fp32 A^T.B and column sums of
    the bf16 operands: ``one_launch`` = qavit_gemm_tn_grouped_ws (all classes in gemm_tn_uni_kernel), ``class_launches`` =
    qavit_gemm_tn_grouped without a workspace (a launch per class and 24 problems)."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    L = importlib.import_module("qa-vit_amd.lib")
    g = torch.Generator().manual_seed(77)
    shapes = [(1500, 256, 1024, False), (1500, 1024, 256, True), (900, 768, 192, False), (900, 192, 768, False), (2100, 64, 256, False),
              (2100, 256, 64, True), (700, 512, 128, True), (700, 128, 512, False), (650, 384, 192, False), (333, 288, 64, False),
              (1000, 256, 256, True), (520, 576, 192, False), (130, 16, 32, False), (4100, 96, 192, True)]
    shapes += [(700 + 8 * i, 192, 192, i % 3 == 0) for i in range(60)]
    probs = []
    for (M, N, Kd, ln) in shapes:
        A = (torch.randn(M, N, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
        Bm = torch.randn(M, Kd, generator=g).to(DEV).to(torch.bfloat16)
        Cg = torch.randn(N, Kd, generator=g).to(DEV)            # the kernels ACCUMULATE into the gradient
        cs = torch.randn(N, generator=g).to(DEV)
        lnp = None
        Bref = Bm.float()
        if ln:
            gam, bet = torch.randn(Kd, generator=g).to(DEV), torch.randn(Kd, generator=g).to(DEV)
            mean = Bref.mean(1).contiguous()
            rstd = (Bref.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
            lnp = (gam, bet, mean, rstd)
            Bref = ((Bref - mean[:, None]) * rstd[:, None] * gam + bet).to(torch.bfloat16).float()   # the kernel rounds the normalised operand to bf16
        probs.append((A, Bm, Cg, cs, lnp, Cg.clone() + A.float().t() @ Bref, cs.clone() + A.float().sum(0)))
    K.DeferredTN.enabled = True
    try:
        for (A, Bm, Cg, cs, lnp, _, _) in probs:
            M, N = A.shape
            K.gemm_tn(A, Bm, Cg, M, N, Bm.shape[1], N, Bm.shape[1], Bm.shape[1], colsum=cs, ln=lnp)
        assert len(K.DeferredTN.queue) == len(probs)
        if mode == "one_launch":
            K.DeferredTN.flush()
        else:
            arr = (L.GemmTnArgs * len(probs))(*[a for a, _, _ in K.DeferredTN.queue])
            L.check(L.load().qavit_gemm_tn_grouped(arr, len(probs), K.stream()), "gemm_tn_grouped")
    finally:
        K.DeferredTN.enabled = False
        K.DeferredTN.queue = []
    torch.cuda.synchronize()
    for (A, Bm, Cg, cs, lnp, Cref, csref), shp in zip(probs, shapes):
        assert rel(Cg, Cref) <= 4e-3, shp
        assert rel(cs, csref) <= 4e-3, shp


@pytest.mark.parametrize("M", [4096, 1000 * 64 + 37])
def test_linear_ln_narrow_output_fused_backward(F, M):
    """TokenLearner's score Linear (LayerNorm -> Linear(192, 16), HQAViT_CIFAR100.py:985-990) with its input also feeding the token mix
    (alias): in bf16 the input-gradient GEMM runs inside the LayerNorm-backward kernel (qavit_layernorm_bwd_lin).  Against fp32 torch
    autograd, and the fused kernel against the GEMM + LayerNorm-backward pair (QAVIT_LN_LIN off)."""
    Kd, N = 192, 16
    x0 = leaf(M, Kd, seed=31)
    w, b = leaf(N, Kd, scale=0.05, seed=32), leaf(N, scale=0.1, seed=33)
    g_, be = leaf(Kd, scale=0.1, seed=34), leaf(Kd, scale=0.1, seed=35)
    with torch.no_grad():
        g_.add_(1.0)
    go = torch.randn(M, N, device=DEV)
    ga = torch.randn(M, Kd, device=DEV)
    res = {}
    for mode in (True, False):
        F._LN_LIN = mode
        for t in (w, b, g_, be):
            t.grad = None
        x = x0.detach().to(torch.bfloat16).requires_grad_(True)
        y, xa = F.linear(x, w, b, ln=(g_, be), alias=True)
        torch.autograd.backward([y, xa], [go.to(torch.bfloat16), ga.to(torch.bfloat16)])
        res[mode] = [t.detach().float().clone() for t in (y, x.grad, w.grad, b.grad, g_.grad, be.grad)]
    F._LN_LIN = True
    xr = x0.detach().to(torch.bfloat16).float().requires_grad_(True)
    wr, br, gr, ber = [t.detach().clone().requires_grad_(True) for t in (w, b, g_, be)]
    yr = TF.linear(TF.layer_norm(xr, (Kd,), gr, ber), wr, br)
    torch.autograd.backward([yr, xr * 1.0], [go.to(torch.bfloat16).float(), ga.to(torch.bfloat16).float()])
    refs = [yr, xr.grad, wr.grad, br.grad, gr.grad, ber.grad]
    for got, ref in zip(res[True], refs):
        assert rel(got, ref) <= tol(torch.bfloat16, False)
    for a_, b_ in zip(res[True], res[False]):
        assert rel(a_, b_) <= 2e-2


@pytest.mark.parametrize("M,Kd,N,act,alias", [(4096, 192, 96, "gelu", True), (1024 + 37, 192, 96, None, False), (70000, 192, 192, None, True),
                                              (20000, 192, 576, None, False), (5000, 128, 512, "gelu", False), (66000, 256, 1024, "gelu", True),
                                              (30000, 128, 96, None, True), (40000, 192, 64, None, True)])
def test_layernorm_backward_in_the_input_gradient_gemm_epilogue(F, Q, M, Kd, N, act, alias):
    """LayerNorm -> Linear (norm2 -> CCF fc1 with GELU, gate_norm -> gate_fc, the ConvNeXt blocks' norm -> pwconv1; HQAViT_CIFAR100.py:704,
    :945, :728): in bf16 the Linear's input-gradient GEMM runs the LayerNorm backward as its epilogue (qavit_gemm_args.e_x: the K-loop
    kernel's column block is the whole LayerNorm row of 128 / 192 / 256) -- dx, the residual alias' gradient added behind it, dgamma /
    dbeta as partial rows or atomics.  Against fp32 torch autograd and against the GEMM + LayerNorm-backward pair (QAVIT_LN_EPILOGUE off);
    every row-tile height (32 / 64 / 128 rows), ragged M, a_mode 0 and 2."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    x0 = leaf(M, Kd, seed=41) * 1.3 + 0.2
    w, b = leaf(N, Kd, scale=0.05, seed=42), leaf(N, scale=0.1, seed=43)
    g_, be = leaf(Kd, scale=0.1, seed=44), leaf(Kd, scale=0.1, seed=45)
    with torch.no_grad():
        g_.add_(1.0)
    go = torch.randn(M, N, device=DEV).to(torch.bfloat16)
    ga = torch.randn(M, Kd, device=DEV).to(torch.bfloat16)
    calls = []
    orig = K.gemm_nt
    K.gemm_nt = lambda *a, **k: (calls.append(k.get("lnbwd") is not None), orig(*a, **k))[1]
    res = {}
    try:
        for mode in (True, False):
            F._LN_EPI = mode
            del calls[:]
            for t in (w, b, g_, be):
                t.grad = None
            x = x0.detach().to(torch.bfloat16).requires_grad_(True)
            if alias:
                y, xa = F.linear(x, w, b, ln=(g_, be), act=act, alias=True)
                torch.autograd.backward([y, xa], [go, ga])
            else:
                y = F.linear(x, w, b, ln=(g_, be), act=act)
                y.backward(go)
            torch.cuda.synchronize()
            assert any(calls) == mode, (mode, calls)            # the epilogue really ran (and only when asked)
            res[mode] = [t.detach().float().clone() for t in (y, x.grad, w.grad, b.grad, g_.grad, be.grad)]
    finally:
        K.gemm_nt = orig
        F._LN_EPI = True
    xr = x0.detach().to(torch.bfloat16).float().requires_grad_(True)
    wr = w.detach().to(torch.bfloat16).float().requires_grad_(True)
    br, gr, ber = [t.detach().clone().requires_grad_(True) for t in (b, g_, be)]
    yr = TF.linear(TF.layer_norm(xr, (Kd,), gr, ber), wr, br)
    if act == "gelu":
        yr = TF.gelu(yr)
    if alias:
        torch.autograd.backward([yr, xr * 1.0], [go.float(), ga.float()])
    else:
        yr.backward(go.float())
    refs = [yr, xr.grad, wr.grad, br.grad, gr.grad, ber.grad]
    for name, got, ref in zip(("y", "dx", "dW", "db", "dgamma", "dbeta"), res[True], refs):
        assert rel(got, ref) <= tol(torch.bfloat16, name == "y"), (name, rel(got, ref))
    for name, a_, b_ in zip(("y", "dx", "dW", "db", "dgamma", "dbeta"), res[True], res[False]):
        assert rel(a_, b_) <= 2e-2, (name, rel(a_, b_))
    assert torch.equal(res[True][0], res[False][0])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_ln_gelu_residual(F, dtype):
    M, K, N = 2048, 192, 96
    x0 = leaf(M, K, seed=5)
    w, b = leaf(N, K, scale=0.05, seed=6), leaf(N, scale=0.1, seed=7)
    g_, be = leaf(K, scale=0.1, seed=8), leaf(K, scale=0.1, seed=9)
    with torch.no_grad():
        g_.add_(1.0)
    res0 = leaf(M, N, seed=10)
    x = x0.detach().to(dtype).requires_grad_(True)
    res = res0.detach().to(dtype).requires_grad_(True)
    refs = [t.detach().clone().float().requires_grad_(True) for t in (x, w, b, g_, be, res)]
    y = F.linear(x, w, b, ln=(g_, be), act="gelu", resid=res)
    xr, wr, br, gr, ber, rr = refs
    ref = rr + TF.gelu(TF.linear(TF.layer_norm(xr, (K,), gr, ber), wr, br))
    assert rel(y, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    y.backward(go.to(dtype))
    ref.backward(go)
    for a, r, name in ((x, xr, "x"), (w, wr, "w"), (b, br, "b"), (g_, gr, "gamma"), (be, ber, "beta"), (res, rr, "res")):
        assert rel(a.grad, r.grad) <= tol(dtype, False) * (3 if name in ("gamma", "beta") else 1), name


def test_linear_rows_slice_and_stack(F):
    M, K = 512, 32
    x = leaf(M, K, seed=11)
    w, b = leaf(48, K, scale=0.1, seed=12), leaf(48, scale=0.1, seed=13)
    y = F.linear(x, w, b, rows=(16, 32))
    ref = TF.linear(x.detach(), w.detach()[16:48], b.detach()[16:48])
    assert rel(y, ref) <= 3e-5
    y.sum().backward()
    assert float(w.grad[:16].abs().max()) == 0.0 and float(w.grad[16:].abs().max()) > 0
    ws = [leaf(16, K, scale=0.1, seed=20 + i) for i in range(3)]
    bs = [leaf(16, scale=0.1, seed=30 + i) for i in range(3)]
    x2 = leaf(M, K, seed=14)
    ys = F.LinearStack3Fn.apply(x2, ws[0], bs[0], ws[1], bs[1], ws[2], bs[2])
    xr = x2.detach().clone().requires_grad_(True)
    wr = [t.detach().clone().requires_grad_(True) for t in ws]
    brr = [t.detach().clone().requires_grad_(True) for t in bs]
    refs = torch.cat([TF.linear(xr, wr[i], brr[i]) for i in range(3)], -1)
    assert rel(ys, refs) <= 3e-5
    go = torch.randn_like(refs)
    ys.backward(go)
    refs.backward(go)
    assert rel(x2.grad, xr.grad) <= 3e-4
    for i in range(3):
        assert rel(ws[i].grad, wr[i].grad) <= 3e-4
        assert rel(bs[i].grad, brr[i].grad) <= 3e-4


def test_dropout_and_droppath_masks(F, Q):
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    M, Kd, N = 4096, 64, 128
    x = leaf(M, Kd, seed=40)
    w = leaf(N, Kd, scale=0.1, seed=41)
    s1, s2 = K.new_site(), K.new_site()
    y = F.linear(x, w, None, drop=(0.25, s1))
    y0 = TF.linear(x.detach(), w.detach())
    kept = (y != 0)
    frac = kept.float().mean().item()
    assert abs(frac - 0.75) < 0.01
    assert rel(y[kept], (y0 / 0.75)[kept]) <= 3e-5
    y.backward(torch.ones_like(y))
    # backward regenerates the same mask: dL/dx = (mask/keep) @ W
    ref_dx = (kept.float() / 0.75) @ w.detach()
    assert rel(x.grad, ref_dx) <= 3e-4
    # drop path: whole samples (rows_per_sample rows) share one factor
    x2 = leaf(M, Kd, seed=42)
    y2 = F.linear(x2, w, None, dp=(0.5, s2, 16))
    y20 = TF.linear(x2.detach(), w.detach()).reshape(M // 16, -1)
    y2r = y2.detach().reshape(M // 16, -1)
    per_sample = torch.round((y2r.abs().sum(1) / y20.abs().sum(1)))
    assert set(per_sample.tolist()) <= {0.0, 2.0}
    assert rel(y2r, y20 * per_sample[:, None]) <= 3e-5
    f = (per_sample > 1).float().mean().item()
    assert 0.35 < f < 0.65
    # a new step gives a new mask
    K.Runtime.get(0).advance()
    y3 = F.linear(x.detach(), w.detach(), None, drop=(0.25, s1))
    assert (y3 != 0).ne(kept).any()


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layernorm(F, dtype):
    B, N, C = 33, 64, 192
    x = leaf(B, N, C, seed=50).detach().to(dtype).requires_grad_(True)
    g_, b = leaf(C, scale=0.2, seed=51), leaf(C, scale=0.2, seed=52)
    pos = leaf(1, N, C, scale=0.3, seed=53)
    y = F.layer_norm(x, g_, b, 1e-5, add=pos)
    xr, gr, br, pr = [t.detach().clone().float().requires_grad_(True) for t in (x, g_, b, pos)]
    ref = TF.layer_norm(xr, (C,), gr, br) + pr
    assert rel(y, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    y.backward(go.to(dtype))
    ref.backward(go)
    assert rel(x.grad, xr.grad) <= tol(dtype, False)
    assert rel(g_.grad, gr.grad) <= tol(dtype, False)
    assert rel(b.grad, br.grad) <= tol(dtype, False)
    assert rel(pos.grad, pr.grad) <= tol(dtype, False)
    # fused exact GELU after the norm (LMFAdapter / SplitFusion.cat_mlp)
    x2 = leaf(B * N, C, seed=54).detach().to(dtype).requires_grad_(True)
    g2, b2 = leaf(C, scale=0.2, seed=55), leaf(C, scale=0.2, seed=56)
    with torch.no_grad():
        g2.add_(1.0)
    y2 = F.layer_norm(x2, g2, b2, 1e-5, act="gelu")
    r2 = [t.detach().clone().float().requires_grad_(True) for t in (x2, g2, b2)]
    ref2 = TF.gelu(TF.layer_norm(r2[0], (C,), r2[1], r2[2]))
    assert rel(y2, ref2) <= tol(dtype)
    go2 = torch.randn_like(ref2)
    y2.backward(go2.to(dtype))
    ref2.backward(go2)
    for a_, r_ in zip((x2, g2, b2), r2):
        assert rel(a_.grad, r_.grad) <= tol(dtype, False) * 2


# ---------------------------------------------------------------------------------------------------
def _ref_attn(q, k, v, keep=None, p=0.0):
    """SDPA; with ``keep`` (bool [G,H,Nq,NK], the mask the kernel used) the reference's dropout_p branch:
    softmax(S) * keep / (1 - p) @ v  (F.scaled_dot_product_attention's documented math)."""
    if keep is None:
        return TF.scaled_dot_product_attention(q, k, v)
    pr = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(q.shape[-1]), -1)
    return (pr * keep.to(pr.dtype) / (1.0 - p)) @ v


def _drop_setup(drop, G, H, Nq, NK):
    """-> (spec entry, keep mask on DEV or None) for the CURRENT (seed, step) of the runtime."""
    if drop <= 0.0:
        return None, None
    import importlib
    from conftest import attn_keep_mask
    K = importlib.import_module("qa-vit_amd.kernels")
    site = K.new_site()
    seed, step = [int(v) for v in K.Runtime.get(0).rng.tolist()]
    keep = torch.from_numpy(attn_keep_mask(seed, step, site, G, H, Nq, NK, drop)).to(DEV)
    frac = float(keep.float().mean())
    assert abs(frac - (1.0 - drop)) < 0.02, frac
    return (drop, site), keep


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N,M", [(5, 64, 16), (700, 64, 16), (300, 256, 64)])
def test_upmix_with_the_block_tail_scale_add(F, Q, dtype, B, N, M):
    """UpMixScaleAddFn = TokenUpMix(x + droppath(gamma * u)) as one node.  bf16: its forward is ONE launch that forms xc while it stages
    the image (qavit_upmix_fwd_sa, both token shapes: y bit-equal to the two launches), its backward (64 -> 16 tokens) ONE launch that also
    writes du and dgamma (qavit_upmix_bwd_sa): against the two separate nodes (ScaleAddFn, then UpMixFn) on the same inputs and masks."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    rt = K.Runtime.get(0)
    C = 192
    x0, u0 = leaf(B, M, C, seed=71), leaf(B, M, C, seed=72)
    W, bias = leaf(N, M, scale=0.3, seed=73), leaf(N, scale=0.1, seed=74)
    g_, be = leaf(C, scale=0.1, seed=75), leaf(C, scale=0.1, seed=76)
    gam = torch.full((1,), 0.3, device=DEV, requires_grad=True)
    with torch.no_grad():
        g_.add_(1.0)
    gy = torch.randn(B, N, C, device=DEV).to(dtype)
    dp = (0.25, 4242, M)
    res = []
    for fused in (True, False):
        rt.seed(99)
        for t in (W, bias, g_, be, gam):
            t.grad = None
        x = x0.detach().to(dtype).requires_grad_(True)
        u = u0.detach().to(dtype).requires_grad_(True)
        if fused:
            y = F.UpMixScaleAddFn.apply(x, u, gam, dp, W, bias, g_, be, 1e-5)
        else:
            y = F.UpMixFn.apply(F.ScaleAddFn.apply(x, u, gam, dp), W, bias, g_, be, 1e-5)
        y.backward(gy)
        res.append([t.detach().float().clone() for t in (y, x.grad, u.grad, gam.grad, W.grad, bias.grad, g_.grad, be.grad)])
    for name, a_, b_ in zip(("y", "dx", "du", "dgamma", "dW", "dbias", "dg", "db"), res[0], res[1]):
        assert rel(a_, b_) <= (2e-5 if dtype == torch.float32 else 4e-3), name
    assert float(res[0][2].abs().max()) > 0 and float(res[0][3].abs().max()) > 0
    if dtype == torch.bfloat16:
        assert K.upmix_fwd_sa_ok(x, u, N, M, C)
        assert torch.equal(res[0][0], res[1][0])               # same arithmetic, same rounding points


@pytest.mark.parametrize("M,C,N", [(4096 + 17, 192, 192), (1024, 128, 256), (70000, 192, 192)])
def test_linear_on_cat_as_one_two_source_gemm(F, Q, M, C, N):
    """LinearCatFn: cat([T, R]) @ W^T + b as ONE GEMM whose A operand switches source at column C (qavit_gemm_args.A2), SplitFusion's
    Linear(2C -> C) (HQAViT_CIFAR100.py:951): against fp32 torch on the bf16-rounded operands (y and every gradient) and against the
    GEMM + accumulating-GEMM pair it replaces; shapes the K-loop kernel does not take are refused, not rerouted."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    dtype = torch.bfloat16
    t0, r0 = leaf(M, C, seed=101), leaf(M, C, seed=102)
    W, b = leaf(N, 2 * C, scale=0.08, seed=103), leaf(N, scale=0.1, seed=104)
    gy = torch.randn(M, N, device=DEV).to(dtype)
    res = []
    for one in (True, False):
        W.grad = None; b.grad = None
        t, r = t0.detach().to(dtype).requires_grad_(True), r0.detach().to(dtype).requires_grad_(True)
        assert F.linear_cat_ok(t, r, W)
        if one:
            y = F.LinearCatFn.apply(t, r, W, b)
        else:
            y = F.linear(r, W, None, cols=(C, C), resid=F.linear(t, W, b, cols=(0, C)))
        y.backward(gy)
        res.append([v.detach().float().clone() for v in (y, t.grad, r.grad, W.grad, b.grad)])
    tr, rr = t0.detach().to(dtype).float().requires_grad_(True), r0.detach().to(dtype).float().requires_grad_(True)
    Wr, br = W.detach().to(dtype).float().requires_grad_(True), b.detach().clone().requires_grad_(True)
    yr = torch.cat([tr, rr], -1) @ Wr.t() + br
    yr.backward(gy.float())
    for name, x_, y_ in zip(("y", "dT", "dR", "dW", "db"), res[0], (yr, tr.grad, rr.grad, Wr.grad, br.grad)):
        assert rel(x_, y_) <= tol(dtype, name == "y"), name
    for name, x_, y_ in zip(("y", "dT", "dR", "dW", "db"), res[0], res[1]):
        assert rel(x_, y_) <= (1.2e-2 if name == "y" else 1e-5), name            # y: one rounding instead of two; the backward is the same launches
    # refused, loudly: a split off the 64-column chunk grid, fp32 operands
    a = torch.randn(2048, 96, device=DEV).to(dtype)
    y = torch.empty(2048, 192, device=DEV, dtype=dtype)
    Wb = torch.randn(192, 192, device=DEV).to(dtype)
    with pytest.raises(RuntimeError):
        K.gemm_nt(a, Wb, y, 2048, 192, 192, 96, 192, 192, None, A2=a, lda2=96, a2_k0=96)
    assert not F.linear_cat_ok(t0.detach(), r0.detach(), W)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C,p", [(64 * 37 + 3, 192, 0.1), (4099, 192, 0.0), (777, 64, 0.1), (1000, 256, 0.1)])
def test_blend_and_final_norm_one_launch(F, Q, dtype, rows, C, p):
    """Mix3LayerNormFn = LayerNorm(s0*a + s1*(t + dropout(h))), SplitFusion's closing pair (HQAViT_CIFAR100.py:953-965), one launch each way
    (qavit_mix3_ln_fwd / _bwd): against Mix3Fn followed by the LayerNorm node on the same inputs and masks -- y, da, dt, dh bit-equal (same
    arithmetic, same rounding points), the parameter gradients to summation-order tolerance -- and against fp32 torch autograd."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    rt = K.Runtime.get(0)
    a0, t0, h0 = leaf(rows, C, seed=81), leaf(rows, C, seed=82), leaf(rows, C, seed=83)
    fw = torch.tensor([0.75, 0.25], device=DEV, requires_grad=True)
    g_, be = leaf(C, scale=0.1, seed=84), leaf(C, scale=0.1, seed=85)
    with torch.no_grad():
        g_.add_(1.0)
    gy = torch.randn(rows, C, device=DEV).to(dtype)
    drop = (p, 9191)
    res = []
    for fused in (True, False):
        rt.seed(77)
        for t in (fw, g_, be):
            t.grad = None
        a, t, h = (v.detach().to(dtype).requires_grad_(True) for v in (a0, t0, h0))
        assert K.mix3_ln_ok(a, t, h, C)
        if fused:
            y = F.Mix3LayerNormFn.apply(a, t, h, fw, drop, g_, be, 1e-5)
        else:
            y = F.layer_norm(F.Mix3Fn.apply(a, t, h, fw, drop), g_, be, 1e-5)
        y.backward(gy)
        res.append([v.detach().float().clone() for v in (y, a.grad, t.grad, h.grad, fw.grad, g_.grad, be.grad)])
    for name, x_, y_ in zip(("y", "da", "dt", "dh"), res[0], res[1]):
        assert torch.equal(x_, y_), name
    for name, x_, y_ in zip(("dfw", "dgamma", "dbeta"), res[0][4:], res[1][4:]):
        assert rel(x_, y_) <= 2e-4, name
    # fp32 torch autograd with the kernel's mask (recovered from the fused node's dh / dt: dh = dt * mask / (1 - p))
    keep = torch.ones(rows, C, device=DEV)
    if p > 0:
        dt_, dh_ = res[0][2], res[0][3]
        keep = torch.where(dt_ != 0, (dh_ / dt_ * (1 - p)).round(), torch.ones_like(dt_))
        frac = float(keep.mean())
        assert abs(frac - (1 - p)) < 0.02, frac
    ar, tr, hr = (v.detach().to(dtype).float().requires_grad_(True) for v in (a0, t0, h0))
    fwr, gr, ber = (v.detach().clone().requires_grad_(True) for v in (fw, g_, be))
    w = torch.softmax(fwr, 0)
    yr = TF.layer_norm(w[0] * ar + w[1] * (tr + hr * keep / (1 - p)), (C,), gr, ber)
    yr.backward(gy.float())
    assert rel(res[0][0], yr) <= tol(dtype)
    for name, x_, y_ in zip(("da", "dt", "dh", "dfw", "dgamma", "dbeta"), res[0][1:], (ar.grad, tr.grad, hr.grad, fwr.grad, gr.grad, ber.grad)):
        assert rel(x_, y_) <= (tol(dtype, False) if name != "dfw" else 10 * tol(dtype, False)), name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows,C,p", [(64 * 37 + 3, 192, 0.1), (4099, 192, 0.0), (33000, 192, 0.1), (777, 64, 0.1)])
def test_gate_blend_and_final_norm_one_launch(F, Q, dtype, rows, C, p):
    """GateMix3LayerNormFn = LayerNorm(s0*(t + sigmoid(g)*r) + s1*(t + dropout(h))), SplitFusion's gate, blend and final norm
    (HQAViT_CIFAR100.py:945-965), one launch each way (qavit_gate_mix3_ln_fwd / _bwd): against GateMixFn -> Mix3Fn -> LayerNorm on the same
    inputs and masks.  y, dr, dg, dh bit-equal; the single t gradient against the two the separate nodes return, added; parameter gradients
    to summation-order tolerance."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    rt = K.Runtime.get(0)
    t0, r0, gl0, h0 = leaf(rows, C, seed=91), leaf(rows, C, seed=92), leaf(rows, C, seed=93), leaf(rows, C, seed=94)
    fw = torch.tensor([0.6, -0.2], device=DEV, requires_grad=True)
    g_, be = leaf(C, scale=0.1, seed=95), leaf(C, scale=0.1, seed=96)
    with torch.no_grad():
        g_.add_(1.0)
    gy = torch.randn(rows, C, device=DEV).to(dtype)
    drop = (p, 777)
    res = []
    for fused in (True, False):
        rt.seed(78)
        for v in (fw, g_, be):
            v.grad = None
        t, r, gl, h = (v.detach().to(dtype).requires_grad_(True) for v in (t0, r0, gl0, h0))
        if fused:
            y = F.GateMix3LayerNormFn.apply(t, r, gl, h, fw, drop, g_, be, 1e-5)
        else:
            y = F.layer_norm(F.Mix3Fn.apply(F.GateMixFn.apply(t, r, gl), t, h, fw, drop), g_, be, 1e-5)
        y.backward(gy)
        res.append([v.detach().float().clone() for v in (y, r.grad, gl.grad, h.grad, t.grad, fw.grad, g_.grad, be.grad)])
    for name, x_, y_ in zip(("y", "dr", "dg", "dh"), res[0], res[1]):
        assert torch.equal(x_, y_), name
    assert rel(res[0][4], res[1][4]) <= (1e-6 if dtype == torch.float32 else 8e-3), "dt"
    for name, x_, y_ in zip(("dfw", "dgamma", "dbeta"), res[0][5:], res[1][5:]):
        assert rel(x_, y_) <= 2e-4, name
    assert float(res[0][1].abs().max()) > 0 and float(res[0][2].abs().max()) > 0 and float(res[0][5].abs().max()) > 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("rows", [16 * 130, 64 * 1000 + 5])
def test_layernorm_fan_out_backward_sums_on_load(F, dtype, rows):
    """LayerNormFanFn: a LayerNorm output with five consumers + the residual alias of its input; backward = ONE launch that sums the
    five gradients (one of them absent) and the residual's on load (qavit_layernorm_bwd_sum), against fp32 torch autograd."""
    C = 192
    x0 = leaf(rows, C, seed=61)
    g_, be = leaf(C, scale=0.1, seed=62), leaf(C, scale=0.1, seed=63)
    with torch.no_grad():
        g_.add_(1.0)
    x = x0.detach().to(dtype).requires_grad_(True)
    outs = F.LayerNormFanFn.apply(x, g_, be, 1e-5, 5)
    ys, xa = outs[:5], outs[5]
    gs = [torch.randn(rows, C, device=DEV).to(dtype) for _ in range(5)]
    torch.autograd.backward([ys[0], ys[1], ys[3], ys[4], xa], [gs[0], gs[1], gs[3], gs[4], gs[2]])     # ys[2] has no consumer
    xr = x0.detach().to(dtype).float().requires_grad_(True)
    gr, ber = g_.detach().clone().requires_grad_(True), be.detach().clone().requires_grad_(True)
    yr = TF.layer_norm(xr, (C,), gr, ber)
    assert rel(ys[0], yr) <= tol(dtype)
    torch.autograd.backward([yr, xr * 1.0], [gs[0].float() + gs[1].float() + gs[3].float() + gs[4].float(), gs[2].float()])
    assert rel(x.grad, xr.grad) <= tol(dtype, False)
    assert rel(g_.grad, gr.grad) <= tol(dtype, False)
    assert rel(be.grad, ber.grad) <= tol(dtype, False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop", [0.0, 0.1])
@pytest.mark.parametrize("B,Hs,ws,KC", [(37, 4, 4, 32), (9, 8, 4, 32), (5, 14, 7, 64), (700, 4, 4, 32)])
def test_attn_swa_like(F, Q, dtype, B, Hs, ws, KC, drop):
    """mode 0 with the window table: qkv [B*N,3C] -> windows -> Linformer(16->32) + 16 bank rows.
    (5, 14, 7, 64): the 224-px windows -- 49 tokens, 64 Linformer rows (80 keys with the bank).
    (700, 4, 4, 32): 2800 (group, head) problems > the 2048-workgroup cap, so a wave visits several problems and its
    dE / bank-gradient partials accumulate across them (the benchmark's regime)."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    N, C, H, S = Hs * Hs, 192, 4, 16
    D = C // H
    nw = Hs // ws
    qkv = leaf(B * N, 3 * C, seed=60).detach().to(dtype).requires_grad_(True)
    Ek, Ev = leaf(ws * ws, KC, scale=0.3, seed=61), leaf(ws * ws, KC, scale=0.3, seed=62)
    bk, bv = leaf(1, S, C, scale=0.5, seed=63), leaf(1, S, C, scale=0.5, seed=64)
    tbl = None
    if nw > 1:
        t = [(wy * ws + ty) * Hs + wx * ws + tx for wy in range(nw) for wx in range(nw) for ty in range(ws) for tx in range(ws)]
        tbl = torch.tensor(t, dtype=torch.int32, device=DEV)
    spec = dict(mode=0, G=B * nw * nw, Nq=ws * ws, L=ws * ws, H=H, D=D, KC=KC, S=S, groups_per_b=nw * nw, q_rows_per_b=N,
                k_rows_per_b=N, q_tbl=tbl, k_tbl=tbl, q_off=0, k_off=C, v_off=2 * C, q_rows=B * N)
    dspec, keep = _drop_setup(drop, B * nw * nw, H, ws * ws, KC + S)
    if dspec:
        spec["drop"] = dspec
    o = F.AttnFn.apply(qkv, None, Ek, Ev, bk, bv, spec)
    # reference (same math as oracle.swa without the projections)
    r = [t.detach().clone().float().requires_grad_(True) for t in (qkv, Ek, Ev, bk, bv)]
    qr, Ekr, Evr, bkr, bvr = r
    x = qr.view(B, Hs, Hs, 3 * C).view(B, nw, ws, nw, ws, 3 * C).permute(0, 1, 3, 2, 4, 5).reshape(-1, ws * ws, 3, H, D).permute(2, 0, 3, 1, 4)
    q, k, v = x[0], x[1], x[2]
    BW = q.shape[0]
    kc = torch.matmul(Ekr.T, k.reshape(BW * H, ws * ws, D)).reshape(BW, H, KC, D)
    vc = torch.matmul(Evr.T, v.reshape(BW * H, ws * ws, D)).reshape(BW, H, KC, D)
    kb = bkr.expand(BW, -1, -1).reshape(BW, S, H, D).transpose(1, 2)
    vb = bvr.expand(BW, -1, -1).reshape(BW, S, H, D).transpose(1, 2)
    ro = _ref_attn(q, torch.cat([kc, kb], 2), torch.cat([vc, vb], 2), keep, drop).transpose(1, 2).reshape(BW, ws * ws, C)
    ro = ro.view(B, nw, nw, ws, ws, C).permute(0, 1, 3, 2, 4, 5).reshape(B * N, C)
    assert rel(o, ro) <= tol(dtype)
    go = torch.randn_like(ro)
    o.backward(go.to(dtype))
    ro.backward(go)
    assert rel(qkv.grad, qr.grad) <= tol(dtype, False)
    for a, rr, nme in ((Ek, Ekr, "Ek"), (Ev, Evr, "Ev"), (bk, bkr, "bank_k"), (bv, bvr, "bank_v")):
        assert rel(a.grad, rr.grad) <= tol(dtype, False), nme


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop", [0.0, 0.1])
@pytest.mark.parametrize("N,NP,KC,B", [(16, 10, 32, 21), (64, 40, 32, 21), (196, 135, 64, 21), (64, 40, 32, 600), (16, 10, 32, 600)])
def test_attn_msda_like(F, dtype, N, NP, KC, B, drop):
    """mode 0, separate q and kv matrices, ragged L (10 / 40 of a 128-row Linformer).  (196, 135, 64) are the 224-px
    dimensions: only the first 128 landmarks are keys (the rest get zero gradient) and the backward keeps E in global
    memory (attn.hip spill layout).  B = 600: 2400 problems > the 2048-workgroup cap (several problems per workgroup)."""
    C, H, S = 192, 4, 16
    D = C // H
    Lk = min(NP, 128)
    q_t = leaf(B * N, C, seed=70).detach().to(dtype).requires_grad_(True)
    kv_t = leaf(B * NP, 2 * C, seed=71).detach().to(dtype).requires_grad_(True)
    Ek, Ev = leaf(128, KC, scale=0.3, seed=72), leaf(128, KC, scale=0.3, seed=73)
    bk, bv = leaf(1, S, C, scale=0.5, seed=74), leaf(1, S, C, scale=0.5, seed=75)
    spec = dict(mode=0, G=B, Nq=N, L=Lk, H=H, D=D, KC=KC, S=S, groups_per_b=1, q_rows_per_b=N, k_rows_per_b=NP,
                q_off=0, k_off=0, v_off=C, q_rows=B * N)
    dspec, keep = _drop_setup(drop, B, H, N, KC + S)
    if dspec:
        spec["drop"] = dspec
    o = F.AttnFn.apply(q_t, kv_t, Ek, Ev, bk, bv, spec)
    qr, kvr, Ekr, Evr, bkr, bvr = [t.detach().clone().float().requires_grad_(True) for t in (q_t, kv_t, Ek, Ev, bk, bv)]
    q = qr.view(B, N, H, D).transpose(1, 2)
    kv = kvr.view(B, NP, 2, H, D).permute(2, 0, 3, 1, 4)
    k = TF.pad(kv[0][:, :, :Lk], (0, 0, 0, 128 - Lk))
    v = TF.pad(kv[1][:, :, :Lk], (0, 0, 0, 128 - Lk))
    kc = torch.matmul(Ekr.T, k.reshape(B * H, 128, D)).reshape(B, H, KC, D)
    vc = torch.matmul(Evr.T, v.reshape(B * H, 128, D)).reshape(B, H, KC, D)
    kb = bkr.expand(B, -1, -1).reshape(B, S, H, D).transpose(1, 2)
    vb = bvr.expand(B, -1, -1).reshape(B, S, H, D).transpose(1, 2)
    ro = _ref_attn(q, torch.cat([kc, kb], 2), torch.cat([vc, vb], 2), keep, drop).transpose(1, 2).reshape(B * N, C)
    assert rel(o, ro) <= tol(dtype)
    go = torch.randn_like(ro)
    o.backward(go.to(dtype))
    ro.backward(go)
    assert rel(q_t.grad, qr.grad) <= tol(dtype, False)
    assert rel(kv_t.grad, kvr.grad) <= tol(dtype, False)
    assert rel(Ek.grad, Ekr.grad) <= tol(dtype, False)
    assert rel(Ev.grad, Evr.grad) <= tol(dtype, False)
    if Lk < 128:
        assert float(Ek.grad[Lk:].abs().max()) == 0.0      # zero-padded rows get no gradient
    if NP > Lk:
        assert float(kv_t.grad.view(B, NP, -1)[:, Lk:].abs().max()) == 0.0      # landmarks past the Linformer length are not keys
    assert rel(bk.grad, bkr.grad) <= tol(dtype, False)
    assert rel(bv.grad, bvr.grad) <= tol(dtype, False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("drop", [0.0, 0.1])
@pytest.mark.parametrize("N,B", [(16, 13), (64, 13), (196, 13), (64, 120)])
def test_attn_cga_like(F, dtype, N, B, drop):
    """mode 1, D=4, keys = own tokens + 16 shared rows, channel-group row table.  B = 120: 2880 problems > the cap."""
    G, H, S, ccg = 6, 4, 16, 16
    D = ccg // H
    qkv = leaf(B * N * G, 3 * ccg, seed=80).detach().to(dtype).requires_grad_(True)
    shk = (leaf(S, ccg, seed=81) * 1.0)
    shv = (leaf(S, ccg, seed=82) * 1.0)
    shk_l, shv_l = shk.detach().clone().requires_grad_(True), shv.detach().clone().requires_grad_(True)
    tbl = torch.tensor([n * G + g for g in range(G) for n in range(N)], dtype=torch.int32, device=DEV)
    spec = dict(mode=1, G=B * G, Nq=N, L=N, H=H, D=D, S=S, groups_per_b=G, q_rows_per_b=N * G, k_rows_per_b=N * G,
                q_tbl=tbl, k_tbl=tbl, q_off=0, k_off=ccg, v_off=2 * ccg, q_rows=B * N * G)
    # non-leaf shared rows (as in the model: outputs of a Linear on the bank)
    sk, sv = shk_l * 1.0, shv_l * 1.0
    dspec, keep = _drop_setup(drop, B * G, H, N, N + S)
    if dspec:
        spec["drop"] = dspec
    o = F.AttnFn.apply(qkv, None, None, None, sk, sv, spec)
    qr = qkv.detach().clone().float().requires_grad_(True)
    skr, svr = shk.detach().clone().requires_grad_(True), shv.detach().clone().requires_grad_(True)
    x = qr.view(B, N, G, 3, H, D).permute(3, 0, 2, 4, 1, 5).reshape(3, B * G, H, N, D)
    kb = skr.view(1, S, H, D).transpose(1, 2).expand(B * G, -1, -1, -1)
    vb = svr.view(1, S, H, D).transpose(1, 2).expand(B * G, -1, -1, -1)
    ro = _ref_attn(x[0], torch.cat([x[1], kb], 2), torch.cat([x[2], vb], 2), keep, drop)       # [BG,H,N,D]
    ro = ro.transpose(1, 2).reshape(B, G, N, ccg).permute(0, 2, 1, 3).reshape(B * N * G, ccg)
    assert rel(o, ro) <= tol(dtype)
    go = torch.randn_like(ro)
    o.backward(go.to(dtype))
    ro.backward(go)
    assert rel(qkv.grad, qr.grad) <= tol(dtype, False)
    assert rel(shk_l.grad, skr.grad) <= tol(dtype, False)
    assert rel(shv_l.grad, svr.grad) <= tol(dtype, False)


@pytest.mark.parametrize("drop", [0.0, 0.1])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,N", [(19, 16), (19, 64), (700, 16)])
def test_attn_cross_like_and_nan_guard(F, dtype, drop, B, N):
    C, H, S = 192, 4, 16
    D = C // H
    q_t = leaf(B * N, C, seed=90).detach().to(dtype).requires_grad_(True)
    shk, shv = leaf(S, C, seed=91), leaf(S, C, seed=92)
    spec = dict(mode=1, G=B, Nq=N, L=0, H=H, D=D, S=S, q_off=0, k_off=0, v_off=0, q_rows=B * N)
    dspec, keep = _drop_setup(drop, B, H, N, S)
    if dspec:
        spec["drop"] = dspec
    o = F.AttnFn.apply(q_t, None, None, None, shk, shv, spec)
    qr, kr, vr = [t.detach().clone().float().requires_grad_(True) for t in (q_t, shk, shv)]
    q = qr.view(B, N, H, D).transpose(1, 2)
    k = kr.view(1, S, H, D).transpose(1, 2).expand(B, -1, -1, -1)
    v = vr.view(1, S, H, D).transpose(1, 2).expand(B, -1, -1, -1)
    ro = _ref_attn(q, k, v, keep, drop).transpose(1, 2).reshape(B * N, C)
    assert rel(o, ro) <= tol(dtype)
    go = torch.randn_like(ro)
    o.backward(go.to(dtype))
    ro.backward(go)
    assert rel(q_t.grad, qr.grad) <= tol(dtype, False)
    assert rel(shk.grad, kr.grad) <= tol(dtype, False)
    assert rel(shv.grad, vr.grad) <= tol(dtype, False)
    if drop > 0.0:
        # a new step draws a new mask; the same step replays the same one
        import importlib
        K = importlib.import_module("qa-vit_amd.kernels")
        o_same = F.AttnFn.apply(q_t.detach(), None, None, None, shk.detach(), shv.detach(), spec)
        assert torch.equal(o_same, o.detach())
        K.Runtime.get(0).advance()
        o_new = F.AttnFn.apply(q_t.detach(), None, None, None, shk.detach(), shv.detach(), spec)
        assert not torch.equal(o_new, o.detach())
        return
    # efficient_attention: any NaN in the inputs -> the WHOLE output is zeros (HQAViT_CIFAR100.py:356-357)
    bad = q_t.detach().clone()
    bad[5, 7] = float("nan")
    ob = F.AttnFn.apply(bad, None, None, None, shk.detach(), shv.detach(), spec)
    assert float(ob.float().abs().max()) == 0.0
    ok = F.AttnFn.apply(q_t.detach(), None, None, None, shk.detach(), shv.detach(), spec)   # flag was cleared
    assert rel(ok, ro) <= tol(dtype)


# ---------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("N,M,B", [(64, 16, 11), (256, 64, 11), (64, 16, 700), (256, 64, 530)])
def test_tokmix_upmix(F, dtype, N, M, B):
    """B = 700 / 530: more images than the up-mix backward's 512 workgroups, so its dW / dgamma partials span images."""
    C = 192
    scores = leaf(B, N, M, seed=100).detach().to(dtype).requires_grad_(True)
    x = leaf(B, N, C, seed=101).detach().to(dtype).requires_grad_(True)
    xc = F.TokMixFn.apply(scores, x)
    sr, xr = [t.detach().clone().float().requires_grad_(True) for t in (scores, x)]
    ref = torch.bmm(torch.softmax(sr, 1).transpose(1, 2), xr)
    assert rel(xc, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    xc.backward(go.to(dtype))
    ref.backward(go)
    assert rel(x.grad, xr.grad) <= tol(dtype, False)
    assert rel(scores.grad, sr.grad) <= tol(dtype, False)
    # up-mix
    xc2 = leaf(B, M, C, seed=102).detach().to(dtype).requires_grad_(True)
    W, bias = leaf(N, M, scale=0.2, seed=103), leaf(N, scale=0.2, seed=104)
    g_, b = leaf(C, scale=0.2, seed=105), leaf(C, scale=0.2, seed=106)
    with torch.no_grad():
        g_.add_(1.0)
    y = F.UpMixFn.apply(xc2, W, bias, g_, b, 1e-5)
    r = [t.detach().clone().float().requires_grad_(True) for t in (xc2, W, bias, g_, b)]
    up = TF.linear(r[0].transpose(1, 2), r[1], r[2]).transpose(1, 2)
    ref = TF.layer_norm(up, (C,), r[3], r[4])
    assert rel(y, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    y.backward(go.to(dtype))
    ref.backward(go)
    for a, rr, nme in zip((xc2, W, bias, g_, b), r, ("xc", "W", "bias", "gamma", "beta")):
        if nme == "bias":      # d/dbias of LN(up + bias) is identically 0 (LN removes the row mean): both are round-off
            assert float(a.grad.abs().max()) <= 1e-3 * float(W.grad.abs().max())
            continue
        assert rel(a.grad, rr.grad) <= tol(dtype, False) * 2, nme


@pytest.mark.parametrize("B", [3, 260, 1500])
def test_fused_token_learner(F, Q, B):
    """TokenLearnerFn = softmax_N(Linear(LayerNorm(x)))^T x (HQAViT_CIFAR100.py:971-1002) as one launch each way (qavit_tl_fwd / qavit_tl_bwd,
    bf16, 64 -> 16 tokens): against fp32 torch autograd on the bf16-rounded operands -- xc, dx and the four parameter gradients the
    backward kernel leaves as partial rows (score weight / bias, LayerNorm gamma / beta) -- and against the chain it replaces (LayerNorm-
    prologue GEMM + TokMixFn; tokmix_bwd + layernorm_bwd_lin + the deferred weight-gradient GEMM).  B = 260 / 1500: more images than the
    backward's 256 workgroups (partial rows span images), B = 3: fewer."""
    import importlib
    M_ = importlib.import_module("qa-vit_amd.modules")
    N, M, C = 64, 16, 192
    dtype = torch.bfloat16
    tl = M_.TokenLearner(C, M).to(DEV)
    g = torch.Generator().manual_seed(5)
    with torch.no_grad():
        tl.attention[0].weight.copy_(1.0 + 0.2 * torch.randn(C, generator=g))
        tl.attention[0].bias.copy_(0.2 * torch.randn(C, generator=g))
        tl.attention[1].weight.copy_(0.3 * torch.randn(M, C, generator=g))
        tl.attention[1].bias.copy_(0.3 * torch.randn(M, generator=g))
    x0 = (leaf(B, N, C, seed=610).detach() * 1.5 + 0.3).to(dtype)
    go = torch.randn(B, M, C, device=DEV).to(dtype)
    res = []
    for fused in (True, False):
        F._TL_FUSED = fused
        try:
            for p_ in tl.parameters():
                p_.grad = None
            x = x0.clone().requires_grad_(True)
            assert F.tl_ok(x, tl.attention[1].weight) == fused
            xc = tl(x)
            xc.backward(go)
            torch.cuda.synchronize()
        finally:
            F._TL_FUSED = True
        res.append(dict(xc=xc.detach().float(), dx=x.grad.float(), dW=tl.attention[1].weight.grad.clone(), db=tl.attention[1].bias.grad.clone(),
                        dg=tl.attention[0].weight.grad.clone(), dbeta=tl.attention[0].bias.grad.clone()))
    # fp32 torch on the operands the kernels see (bf16 tokens, bf16-rounded score weight)
    xr = x0.float().requires_grad_(True)
    gr, br = (t.detach().clone().requires_grad_(True) for t in (tl.attention[0].weight, tl.attention[0].bias))
    Wr = tl.attention[1].weight.detach().to(dtype).float().requires_grad_(True)
    b2 = tl.attention[1].bias.detach().clone().requires_grad_(True)
    sc = TF.linear(TF.layer_norm(xr, (C,), gr, br), Wr, b2)
    ref = torch.bmm(torch.softmax(sc, 1).transpose(1, 2), xr)
    ref.backward(go.float())
    refs = dict(xc=ref, dx=xr.grad, dW=Wr.grad, db=b2.grad, dg=gr.grad, dbeta=br.grad)
    # d/dbias is identically zero in exact arithmetic (a softmax over the tokens ignores a per-column constant): what either path holds is the
    # sum of its bf16 score-gradient roundings -- bounded against the weight gradient's size and against the unfused path's own residue
    wmax = float(refs["dW"].abs().max())
    assert float(res[0]["db"].abs().max()) <= max(3.0 * float(res[1]["db"].abs().max()), 2e-2 * wmax), (res[0]["db"].abs().max(), res[1]["db"].abs().max(), wmax)
    # ... and so is d/dbeta of the LayerNorm: dbeta = W^T . (column sums of the score gradient) = W^T . 0
    gmax = float(refs["dg"].abs().max())
    assert float(res[0]["dbeta"].abs().max()) <= max(3.0 * float(res[1]["dbeta"].abs().max()), 2e-2 * gmax), (res[0]["dbeta"].abs().max(), res[1]["dbeta"].abs().max(), gmax)
    for k_ in ("xc", "dx", "dW", "dg"):
        t_ = tol(dtype, k_ == "xc")
        assert rel(res[0][k_], refs[k_]) <= t_, (k_, "fused vs fp32 torch", rel(res[0][k_], refs[k_]))
        assert rel(res[1][k_], refs[k_]) <= t_ * 1.5, (k_, "unfused vs fp32 torch")
        assert rel(res[0][k_], res[1][k_]) <= t_, (k_, "fused vs unfused")
    # the forward keeps the unfused chain's rounding points (bf16 LayerNorm output, scores, P); the row statistics are summed in another
    # order (4 lanes x 48 channels here, 8 lanes x 24 in the GEMM prologue), so a few normalised values round the other way: within 2 bf16 ulps
    assert rel(res[0]["xc"], res[1]["xc"]) <= 2.0 ** -6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Hs,dil", [(4, (1, 2)), (8, (1, 2)), (14, (1, 2, 3)), (4, (1, 1, 1, 1, 1, 2))])
def test_gather_pool(F, dtype, Hs, dil):
    """MSDA landmarks: dilated grid gather + AvgPool1d(2,2) over the gathered sequence (HQAViT_CIFAR100.py:493-503).
    The last case repeats dilation 1 five times so every token has more than 4 readers (the backward's scan path)."""
    B, C, stride = 7, 192, 2
    N = Hs * Hs
    t = []
    for d in dil:
        for y in range(0, Hs, d):
            for x_ in range(0, Hs, d):
                t.append(y * Hs + x_)
    t = t[: (len(t) // stride) * stride]
    idx = torch.tensor(t, dtype=torch.int32, device=DEV)
    x = leaf(B, N, C, seed=400).detach().to(dtype).requires_grad_(True)
    y = F.GatherPoolFn.apply(x, idx, stride)
    xr = x.detach().clone().float().requires_grad_(True)
    ref = TF.avg_pool1d(xr[:, idx.long()].transpose(1, 2), stride, stride).transpose(1, 2)
    assert rel(y, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    y.backward(go.to(dtype))
    ref.backward(go)
    assert rel(x.grad, xr.grad) <= tol(dtype, False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Hs,flags", [(4, "hqa"), (8, "hqa"), (8, "v1"), (4, "v2"), (14, "v1"), (14, "v2")])
def test_ccf_mid(F, dtype, Hs, flags):
    """Hs=14 (224 px) exceeds the fused kernel's LDS tile: functional.ccf_mid composes LN / dwconv / LN instead."""
    B, C = 29, 96
    N = Hs * Hs
    h = leaf(B, N, C, seed=110).detach().to(dtype).requires_grad_(True)
    w = leaf(C, 1, 3, 3, scale=0.3, seed=111)
    ln = flags != "v1"
    bias = leaf(C, scale=0.2, seed=112) if flags in ("v1", "v2") else None
    scale = (leaf(1, C, 1, 1, scale=0.05, seed=113).detach() + 0.1).requires_grad_(True) if flags != "v1" else None
    g1, b1, g2, b2 = [leaf(C, scale=0.2, seed=114 + i) if ln else None for i in range(4)]
    out = F.ccf_mid(h, g1, b1, g2, b2, w, bias, scale, Hs, Hs, 1e-5)
    params = [t for t in (h, w, bias, scale, g1, b1, g2, b2)]
    r = [None if t is None else t.detach().clone().float().requires_grad_(True) for t in params]
    hr, wr, br, sr, g1r, b1r, g2r, b2r = r
    a = TF.layer_norm(hr, (C,), g1r, b1r) if ln else hr
    img = TF.conv2d(a.transpose(1, 2).reshape(B, C, Hs, Hs), wr, br, padding=1, groups=C)
    if sr is not None:
        img = img * sr
    t = img.flatten(2).transpose(1, 2)
    ref = TF.layer_norm(t, (C,), g2r, b2r) if ln else t
    assert rel(out, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    out.backward(go.to(dtype))
    ref.backward(go)
    for a_, rr in zip(params, r):
        if a_ is not None:
            assert rel(a_.grad, rr.grad) <= tol(dtype, False) * 2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(7, 64, 8), (7, 256, 8), (3, 128, 8), (5, 100, 8), (7, 64, 16), (3, 100, 16), (5, 192, 16), (3, 128, 24), (7, 64, 12), (3, 64, 16, 700), (7, 64, 8, 1200)])
def test_dwconv_tokens(F, dtype, case):
    """4-tuples carry a batch large enough that a wave visits several (image, tile) units and its tap sums span them."""
    ks, C, H = case[:3]
    B = case[3] if len(case) > 3 else 21
    x = leaf(B, H * H, C, seed=130).detach().to(dtype).requires_grad_(True)
    w = leaf(C, 1, ks, ks, scale=0.2, seed=131)
    b = leaf(C, scale=0.2, seed=132)
    y = F.DwConvFn.apply(x, w, b, H, H)
    xr, wr, br = [t.detach().clone().float().requires_grad_(True) for t in (x, w, b)]
    ref = TF.conv2d(xr.transpose(1, 2).reshape(B, C, H, H), wr, br, padding=ks // 2, groups=C).flatten(2).transpose(1, 2)
    assert rel(y, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    y.backward(go.to(dtype))
    ref.backward(go)
    assert rel(x.grad, xr.grad) <= tol(dtype, False)
    assert rel(w.grad, wr.grad) <= tol(dtype, False)
    assert rel(b.grad, br.grad) <= tol(dtype, False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("p", [0.0, 0.1])
def test_mix3_is_dropout_add_mix2(F, dtype, p):
    """SplitFusion's  softmax(fw) . (a, t + dropout(h))  (HQAViT_CIFAR100.py:953-963) in one kernel each way: the same mask (site,
    flat element index) and the same roundings as the three-launch chain, so values and gradients are identical."""
    shp = (37, 64, 192)
    a, t, h = [leaf(*shp, seed=160 + i).detach().to(dtype).requires_grad_(True) for i in range(3)]
    fw = torch.tensor([0.75, 0.25], device=DEV, requires_grad=True)
    site = 4321
    y = F.Mix3Fn.apply(a, t, h, fw, (p, site))
    a2, t2, h2 = [v.detach().clone().requires_grad_(True) for v in (a, t, h)]
    fw2 = fw.detach().clone().requires_grad_(True)
    y2 = F.Mix2Fn.apply(a2, t2 + F.dropout(h2, p, site, True), fw2)
    assert torch.equal(y, y2)
    go = torch.randn_like(y)
    y.backward(go)
    y2.backward(go)
    for g, r in ((a, a2), (t, t2), (h, h2)):
        assert torch.equal(g.grad, r.grad)
    assert rel(fw.grad, fw2.grad) <= 1e-4
    if p > 0:
        assert float((h.grad == 0).float().mean()) > 0.05          # the mask is applied


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(64, 8, 21), (256, 8, 70), (40, 16, 5)])
def test_lmf_gather(F, dtype, case):
    """LMFAdapter's cat([dw3(x), dw5(x), x]) (HQAViT_CIFAR100.py:830-834) as one node: the convolutions write / read column slices of the
    3C-wide buffer (row strides), the pass-through gradient and the first convolution's dx are added inside the kernels."""
    C, H, B = case
    x = leaf(B, H * H, C, seed=150).detach().to(dtype).requires_grad_(True)
    w3, b3 = leaf(C, 1, 3, 3, scale=0.2, seed=151), leaf(C, scale=0.2, seed=152)
    w5, b5 = leaf(C, 1, 5, 5, scale=0.2, seed=153), leaf(C, scale=0.2, seed=154)
    y = F.LmfGatherFn.apply(x, w3, b3, w5, b5, H, H)
    xr, w3r, b3r, w5r, b5r = [t.detach().clone().float().requires_grad_(True) for t in (x, w3, b3, w5, b5)]
    img = xr.transpose(1, 2).reshape(B, C, H, H)
    ref = torch.cat([TF.conv2d(img, w3r, b3r, padding=1, groups=C), TF.conv2d(img, w5r, b5r, padding=2, groups=C), img], 1).flatten(2).transpose(1, 2)
    assert y.shape == ref.shape
    assert rel(y, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    y.backward(go.to(dtype))
    ref.backward(go)
    assert rel(x.grad, xr.grad) <= tol(dtype, False)
    for g, r in ((w3, w3r), (b3, b3r), (w5, w5r), (b5, b5r)):
        assert rel(g.grad, r.grad) <= tol(dtype, False)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_s2_as_im2col_gemm(F, dtype):
    """Both stem convolutions: NCHW fp32 image source (no input grad) and channel-last token source (col2im grad)."""
    B = 9
    img = torch.randn(B, 3, 32, 32, device=DEV)
    w0, b0 = leaf(32, 3, 3, 3, scale=0.2, seed=140), leaf(32, scale=0.1, seed=141)
    cols = F.Im2ColFn.apply(img, (B, 3, 32, 32, 3, 2, 1), dtype)
    assert cols.shape[1] == 32 and float(cols[:, 27:].abs().max()) == 0.0        # K = 27 padded to 64-byte rows of zeros
    y0 = F.linear(cols, w0, b0, xpad=True).reshape(B, 256, 32)
    w0r, b0r = w0.detach().clone().requires_grad_(True), b0.detach().clone().requires_grad_(True)
    ref0 = TF.conv2d(img, w0r, b0r, stride=2, padding=1).flatten(2).transpose(1, 2)
    assert rel(y0, ref0) <= tol(dtype)
    go0 = torch.randn_like(ref0)
    y0.backward(go0.to(dtype))
    ref0.backward(go0)
    assert rel(w0.grad, w0r.grad) <= tol(dtype, False)
    assert rel(b0.grad, b0r.grad) <= tol(dtype, False)
    t = leaf(B, 256, 32, seed=142).detach().to(dtype).requires_grad_(True)
    w1, b1 = leaf(64, 32, 3, 3, scale=0.1, seed=143), leaf(64, scale=0.1, seed=144)
    y1 = F.linear(F.Im2ColFn.apply(t, (B, 32, 16, 16, 3, 2, 1), dtype), w1, b1).reshape(B, 64, 64)
    tr, wr, br = [v.detach().clone().float().requires_grad_(True) for v in (t, w1, b1)]
    ref1 = TF.conv2d(tr.transpose(1, 2).reshape(B, 32, 16, 16), wr, br, stride=2, padding=1).flatten(2).transpose(1, 2)
    assert rel(y1, ref1) <= tol(dtype)
    go = torch.randn_like(ref1)
    y1.backward(go.to(dtype))
    ref1.backward(go)
    assert rel(t.grad, tr.grad) <= tol(dtype, False)
    assert rel(w1.grad, wr.grad) <= tol(dtype, False)
    assert rel(b1.grad, br.grad) <= tol(dtype, False)


def test_small_ops(F):
    B, N, C = 17, 16, 192
    x = leaf(B, N, C, seed=120)
    fw = leaf(4, scale=0.5, seed=121)
    y = F.HybridFuseFn.apply(x, fw)
    xr, fr = x.detach().clone().requires_grad_(True), fw.detach().clone().requires_grad_(True)
    wsm = torch.softmax(fr, 0)
    ref = torch.cat([xr[..., 48 * i:48 * (i + 1)] * wsm[i] for i in range(4)], -1)
    assert rel(y, ref) <= 3e-5
    go = torch.randn_like(ref)
    y.backward(go)
    ref.backward(go)
    assert rel(x.grad, xr.grad) <= 3e-5 and rel(fw.grad, fr.grad) <= 3e-4
    # scale-add
    u, gamma = leaf(B, N, C, seed=122), leaf(1, scale=0.1, seed=123)
    x2 = leaf(B, N, C, seed=124)
    y2 = F.ScaleAddFn.apply(x2, u, gamma, (0.0, 0, N))
    r = [t.detach().clone().requires_grad_(True) for t in (x2, u, gamma)]
    ref2 = r[0] + r[2] * r[1]
    assert rel(y2, ref2) <= 3e-5
    y2.backward(go)
    ref2.backward(go)
    assert rel(u.grad, r[1].grad) <= 3e-5 and rel(gamma.grad, r[2].grad) <= 3e-4 and rel(x2.grad, r[0].grad) <= 1e-6
    # token mean, gather-pool
    x3 = leaf(B, 64, C, seed=125)
    m = F.TokenMeanFn.apply(x3)
    assert rel(m, x3.detach().mean(1)) <= 3e-5
    idx_l = [y_ * 8 + x_ for d in (1, 2) for y_ in range(0, 8, d) for x_ in range(0, 8, d)]
    idx = torch.tensor(idx_l, dtype=torch.int32, device=DEV)
    pooled = F.GatherPoolFn.apply(x3, idx, 2)
    xr3 = x3.detach().clone().requires_grad_(True)
    refp = TF.avg_pool1d(xr3[:, idx.long()].transpose(1, 2), 2, 2).transpose(1, 2)
    assert rel(pooled, refp) <= 3e-5
    gp = torch.randn_like(refp)
    pooled.backward(gp)
    refp.backward(gp)
    assert rel(x3.grad, xr3.grad) <= 3e-5
    # patchify == unfold of the conv
    img = torch.randn(5, 3, 32, 32, device=DEV)
    cols = F.patchify(img, 4, torch.float32)
    refc = TF.unfold(img, 4, stride=4).transpose(1, 2).reshape(5 * 64, 48)
    assert rel(cols, refc) <= 1e-7


@pytest.mark.parametrize("N", [16, 196])           # 196 = the 224-px token count: chunked two-pass statistics
def test_bank_write_matches_oracle(F, Q, oracle, N):
    cfg = Q.HQAViTConfig()
    bank = Q.HQAViT(cfg).global_bank
    Q.fill_module(bank)
    P = {("global_bank." + k): v.clone() for k, v in bank.state_dict().items()}
    bank = bank.cuda()
    tokens = torch.randn(37, N, 192)
    ng, nb = torch.randn(192) * 0.1 + 1, torch.randn(192) * 0.1
    for step in range(3):
        pre = torch.nn.functional.layer_norm(tokens + step, (192,), ng, nb)
        oracle.bank_write(P, pre, oracle.VARIANTS["hqa"], True)
        F.bank_write((tokens + step).cuda(), ng.cuda(), nb.cuda(), bank, 0)
    assert rel(bank.global_k, P["global_bank.global_k"]) <= 1e-5
    assert rel(bank.global_v, P["global_bank.global_v"]) <= 1e-5
    assert int(bank.update_count) == 3 == int(P["global_bank.update_count"])


@pytest.mark.parametrize("N", [16, 64, 196])
def test_bank_write_bf16_path_close_to_fp32(F, Q, N):
    """The bf16 fast path of the bank statistics (tokens_bf16.hip) against the fp32 kernel on the same tokens."""
    banks = []
    tokens = torch.randn(53, N, 192, device=DEV)
    ng, nb = (torch.randn(192, device=DEV) * 0.1 + 1), torch.randn(192, device=DEV) * 0.1
    for dtype in (torch.float32, torch.bfloat16):
        bank = Q.HQAViT(Q.HQAViTConfig()).global_bank
        Q.fill_module(bank)
        bank = bank.cuda()
        for step in range(3):
            F.bank_write((tokens + step).to(dtype), ng, nb, bank, 0)
        banks.append((bank.global_k.detach().clone(), bank.global_v.detach().clone(), int(bank.update_count)))
    # the bank moves by <= 3 * 0.005 * 0.05; compare the MOVEMENT, not the (much larger) bank itself
    ref = Q.HQAViT(Q.HQAViTConfig()).global_bank
    Q.fill_module(ref)
    k0, v0 = ref.global_k.cuda(), ref.global_v.cuda()
    assert rel(banks[1][0] - k0, banks[0][0] - k0) <= 5e-2
    assert rel(banks[1][1] - v0, banks[0][1] - v0) <= 5e-2
    assert banks[0][2] == banks[1][2] == 3


def test_adamw_and_l2norm_match_torch(Q):
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    n = 100_003
    p0 = torch.randn(n, device=DEV)
    g = torch.randn(n, device=DEV) * 0.01
    p, m, v = p0.clone(), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    pr = p0.clone().requires_grad_(True)
    opt = torch.optim.AdamW([pr], lr=6e-4, weight_decay=0.06, betas=(0.9, 0.999))
    lr = torch.tensor([6e-4], device=DEV)
    step = torch.zeros(1, device=DEV)
    gn, part = torch.zeros(1, device=DEV), torch.zeros(1024, device=DEV)
    for i in range(3):
        step += 1
        K.l2norm(g, part, gn)
        assert abs(gn.item() - g.norm().item()) <= 1e-5 * g.norm().item()
        K.adamw(p, g, m, v, None, lr, 0.9, 0.999, 1e-8, 0.06, step, gn, 0.5)
        pr.grad = g.clone()
        torch.nn.utils.clip_grad_norm_([pr], 0.5)
        opt.step()
    assert rel(p, pr) <= 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("C", [64, 128, 256])
def test_spatial_layernorm_and_layer_scale(F, dtype, C):
    """nn.LayerNorm([C,8,8]) on channel-last tokens and the ConvNeXt layer scale (HQAViTv2_CIFAR100.py:766, :744-748)."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    B, Hs = 19, 8
    N = Hs * Hs
    x = leaf(B, N, C, seed=300).detach().to(dtype).requires_grad_(True)
    w, b = leaf(C, Hs, Hs, scale=0.2, seed=301), leaf(C, Hs, Hs, scale=0.2, seed=302)
    with torch.no_grad():
        w.add_(1.0)
    y = F.SpatialLayerNormFn.apply(x, w, b, 1e-6)
    xr, wr, br = [t.detach().clone().float().requires_grad_(True) for t in (x, w, b)]
    img = xr.transpose(1, 2).reshape(B, C, Hs, Hs)
    ref = TF.layer_norm(img, (C, Hs, Hs), wr, br, 1e-6).flatten(2).transpose(1, 2)
    assert rel(y, ref) <= tol(dtype)
    go = torch.randn_like(ref)
    y.backward(go.to(dtype))
    ref.backward(go)
    assert rel(x.grad, xr.grad) <= tol(dtype, False)
    assert rel(w.grad, wr.grad) <= tol(dtype, False)
    assert rel(b.grad, br.grad) <= tol(dtype, False)
    # layer scale + drop path + residual
    t = leaf(B, N, C, seed=303).detach().to(dtype).requires_grad_(True)
    u = leaf(B, N, C, seed=304).detach().to(dtype).requires_grad_(True)
    gam = leaf(C, scale=0.5, seed=305)
    out = F.ChanScaleAddFn.apply(t, u, gam, (0.0, 0, N))
    tr, ur, gr = [v.detach().clone().float().requires_grad_(True) for v in (t, u, gam)]
    ref2 = tr + gr * ur
    assert rel(out, ref2) <= tol(dtype)
    go2 = torch.randn_like(ref2)
    out.backward(go2.to(dtype))
    ref2.backward(go2)
    assert rel(t.grad, tr.grad) <= tol(dtype, False)
    assert rel(u.grad, ur.grad) <= tol(dtype, False)
    assert rel(gam.grad, gr.grad) <= tol(dtype, False)
    # drop path: whole samples share one factor in {0, 1/keep}; backward regenerates the same mask
    site = K.new_site()
    t2 = t.detach().clone().requires_grad_(True)
    u2 = u.detach().clone().requires_grad_(True)
    o2 = F.ChanScaleAddFn.apply(t2, u2, gam.detach(), (0.5, site, N))
    fac = ((o2.detach().float() - t2.detach().float()) / (gam.detach() * u2.detach().float())).reshape(B, -1).median(1).values
    assert set(torch.round(fac).tolist()) <= {0.0, 2.0}
    o2.backward(torch.ones_like(o2))
    per = (u2.grad.float() / gam.detach()).reshape(B, -1).median(1).values
    assert torch.equal(torch.round(per), torch.round(fac))


def test_local_clip_matches_clip_grad_norm(Q):
    """Per-parameter clip of the stem / depthwise-conv gradients (reference :1416-1418) on segments of the flat buffer."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    gen = torch.Generator().manual_seed(5)
    flat = (torch.randn(40_000, generator=gen) * 0.02).to(DEV)
    segs = [(0, 27), (64, 4704), (4800, 1), (8192, 20_000), (30_016, 3)]        # some above, some below the threshold
    flat[30_016:30_019] = torch.tensor([0.01, 0.0, -0.01], device=DEV)
    ref = flat.clone()
    for o, n in segs:
        t = ref[o:o + n].clone().requires_grad_(True)
        t.grad = ref[o:o + n].clone()
        torch.nn.utils.clip_grad_norm_([t], 0.1)
        ref[o:o + n] = t.grad
    ws = torch.zeros(2 * len(segs), device=DEV)
    K.local_clip(flat, torch.tensor(segs, dtype=torch.int64, device=DEV), 0.1, ws)
    assert rel(flat, ref) <= 1e-6
    assert float(ws.abs().max()) == 0.0                                          # scratch left clean for the next step
    assert torch.equal(flat[30_016:30_019], ref[30_016:30_019])                  # below the threshold: untouched


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cc,gelu", [(32, True), (256, False)])
def test_batchnorm_matches_torch(F, dtype, Cc, gelu):
    """csrc/bnorm.hip against nn.BatchNorm2d semantics (TF.batch_norm, + exact GELU): output, running statistics,
    dx, dgamma, dbeta.  The pivoted sums must survive a mean far from zero."""
    M = 4099
    x = (leaf(M, Cc, seed=300).detach() * 1.7 + 3.0).to(dtype).requires_grad_(True)
    w, b = leaf(Cc, scale=0.3, seed=301), leaf(Cc, scale=0.3, seed=302)
    with torch.no_grad():
        w.add_(1.0)
    rm, rv = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    y = F.BatchNormFn.apply(x, w, b, rm, rv, 0.1, 1e-5, gelu, True)
    xr, wr, br = [t.detach().clone().float().requires_grad_(True) for t in (x, w, b)]
    rmr, rvr = torch.zeros(Cc, device=DEV), torch.ones(Cc, device=DEV)
    ref = TF.batch_norm(xr, rmr, rvr, wr, br, True, 0.1, 1e-5)
    if gelu:
        ref = TF.gelu(ref)
    assert rel(y, ref) <= tol(dtype)
    assert rel(rm, rmr) <= 1e-4 and rel(rv, rvr) <= 1e-4
    go = torch.randn_like(ref)
    y.backward(go.to(dtype))
    ref.backward(go)
    assert rel(x.grad, xr.grad) <= tol(dtype, False) * 2
    assert rel(w.grad, wr.grad) <= tol(dtype, False) * 2
    assert rel(b.grad, br.grad) <= tol(dtype, False) * 2
    # eval mode: running statistics
    ye = F.BatchNormFn.apply(x.detach(), w, b, rm, rv, 0.1, 1e-5, gelu, False)
    re_ = TF.batch_norm(xr.detach(), rmr, rvr, wr, br, False, 0.1, 1e-5)
    assert rel(ye, TF.gelu(re_) if gelu else re_) <= tol(dtype)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_splitfusion_helpers_match_torch(F, dtype):
    """GateMixFn (t + sigmoid(g)*r), Mix2Fn (softmax-weighted blend, scalar-weight gradients through own reduction) and
    FanOutFn (k-way gradient fan-in) against their torch expressions."""
    n, Cc = 4096, 192
    t, r, g_ = [leaf(n, Cc, seed=400 + i).detach().to(dtype).requires_grad_(True) for i in range(3)]
    fw = torch.tensor([0.75, 0.25], device=DEV, requires_grad=True)
    tr_, rr, gr, fr = [v.detach().clone().float().requires_grad_(True) for v in (t, r, g_, fw)]
    t1, t2 = F.FanOutFn.apply(t, 2)
    r1, r2, r3 = F.FanOutFn.apply(r, 3)
    y = F.Mix2Fn.apply(F.GateMixFn.apply(t1, r1, g_), t2 + r2 * 0.5 + r3, fw)
    w = torch.softmax(fr, 0)
    ref = w[0] * (tr_ + torch.sigmoid(gr) * rr) + w[1] * (tr_ + rr * 0.5 + rr)
    assert rel(y, ref) <= tol(dtype)
    # d fw = s0 s1 (sum go*a - sum go*b) is a difference of two sums of 786 k random-sign terms: with a purely random ``go`` it is
    # a few units while the bf16 rounding of a and b moves each sum by sqrt(n) * 2^-9 ~ 2, so the comparison would hinge on the
    # RNG stream.  A component of ``go`` along (a - b) makes the quantity large against that noise.
    gen = torch.Generator(device=DEV).manual_seed(4242)
    with torch.no_grad():
        go = torch.randn(ref.shape, device=DEV, generator=gen) + 0.25 * (torch.sigmoid(gr) - 1.5) * rr
    y.backward(go.to(dtype))
    ref.backward(go)
    for a_, b_, nm in ((t, tr_, "t"), (r, rr, "r"), (g_, gr, "g"), (fw, fr, "fw")):
        assert rel(a_.grad, b_.grad) <= tol(dtype, False) * 2, nm


# ---------------------------------------------------------------------------------------------------
# fused attention branch (csrc/branch_fwd.hip): one launch = qkv GEMM + Linformer + bank + softmax(+dropout) + PV + proj(+dropout)
# ---------------------------------------------------------------------------------------------------
def _branch_reference(kind, x, wqkv, bqkv, wproj, bproj, Ek, Ev, bk, bv, idx, stride, keep, p_attn):
    """The reference chain in fp32 torch (HQAViT_CIFAR100.py:441-469 / :496-532 / :613-626 without the bank write)."""
    B, T, C = x.shape
    H, D = 4, C // 4
    if kind == 0 and T == 64:          # window_partition (HQAViT_IN_Tiny.py:756-769): four 4x4 windows = four problems; window_reverse below
        xw = x.view(B, 2, 4, 2, 4, C).permute(0, 1, 3, 2, 4, 5).reshape(B * 4, 16, C)
        out, o = _branch_reference(kind, xw, wqkv, bqkv, wproj, bproj, Ek, Ev, bk, bv, idx, stride, keep, p_attn)
        rev = lambda t: t.view(B, 2, 2, 4, 4, C).permute(0, 1, 3, 2, 4, 5).reshape(B, 64, C)
        return rev(out), rev(o)
    if kind == 2:
        q = TF.linear(x, wqkv, bqkv).view(B, T, H, D).transpose(1, 2)
        k = bk.view(1, -1, H, D).transpose(1, 2).expand(B, -1, -1, -1)
        v = bv.view(1, -1, H, D).transpose(1, 2).expand(B, -1, -1, -1)
        o = _ref_attn(q, k, v, keep, p_attn)
    else:
        if kind == 0:
            qkv = TF.linear(x, wqkv, bqkv).view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
            q, k, v = qkv[0], qkv[1], qkv[2]
            Lk = T
        else:
            pooled = x[:, idx.long()].view(B, -1, stride, C).mean(2)
            Lk = pooled.shape[1]
            q = TF.linear(x, wqkv[:C], bqkv[:C]).view(B, T, H, D).transpose(1, 2)
            kv = TF.linear(pooled, wqkv[C:], bqkv[C:]).view(B, Lk, 2, H, D).permute(2, 0, 3, 1, 4)
            k, v = kv[0], kv[1]
        kc = torch.matmul(Ek[:Lk].T, k.reshape(B * H, Lk, D)).reshape(B, H, -1, D)
        vc = torch.matmul(Ev[:Lk].T, v.reshape(B * H, Lk, D)).reshape(B, H, -1, D)
        kb = bk.expand(B, -1, -1).reshape(B, -1, H, D).transpose(1, 2)
        vb = bv.expand(B, -1, -1).reshape(B, -1, H, D).transpose(1, 2)
        o = _ref_attn(q, torch.cat([kc, kb], 2), torch.cat([vc, vb], 2), keep, p_attn)
    o = o.transpose(1, 2).reshape(B, T, C)
    return TF.linear(o, wproj, bproj), o


def _msda_idx(T, stride=2, dil=(1, 2)):
    """MSDA's landmark source rows (HQAViT_CIFAR100.py:497-501): dilated grids concatenated, cut to a multiple of the pooling stride."""
    Hs = int(round(T ** 0.5))
    t = [y * Hs + xx for d in dil for y in range(0, Hs, d) for xx in range(0, Hs, d)]
    t = t[: (len(t) // stride) * stride]
    return torch.tensor(t, dtype=torch.int32, device=DEV), len(t) // stride


def _win_tbl(T):
    """Row table of window_partition with window 4 on the sqrt(T) grid (what modules.EfficientSpatialWindowAttention builds)."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    Hs, ws = int(round(T ** 0.5)), 4
    nw = Hs // ws
    return K.Runtime.get(0).table(("win", Hs, ws), lambda: [(wy * ws + ty) * Hs + wx * ws + tx for wy in range(nw) for wx in range(nw) for ty in range(ws) for tx in range(ws)])


def _attn_keep(seed, step, site, kind, B, T, H, NK, drop):
    """The attention-dropout mask of a fused branch in the torch reference's batch order: SWA on 64 tokens = B * 4 window problems of 16
    queries; everything else B problems of T queries."""
    from conftest import attn_keep_mask
    if drop <= 0:
        return None
    G, Nq = (B * (T // 16), 16) if kind == 0 else (B, T)
    return torch.from_numpy(attn_keep_mask(seed, step, site, G, H, Nq, NK, drop)).to(DEV)


@pytest.mark.parametrize("B,T", [(1030, 16), (130, 64)])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_fused_branch_counted_waits_equal_draining_waits(F, Q, kind, B, T):
    """The fused branch forward streams its weights through an LDS ring behind COUNTED s_waitcnt vmcnt(N) waits; the q / k / v / O saves of
    the training form put global stores between the ring's loads, and the wait arithmetic counts those stores by position (three per
    burst).  qavit_branch_args.drain_waits makes every ring step wait for everything instead: the training form's output and every saved
    tensor must be the same BITS either way -- a miscounted burst (a wave reading a ring slot that has not landed) would show here, on
    more tiles than one round of workgroups."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    C, S, KC = 192, 16, 32
    x = leaf(B, T, C, seed=320).detach().to(torch.bfloat16)
    n_qkv = C if kind == 2 else 3 * C
    wqkv, bqkv = leaf(n_qkv, C, scale=0.08, seed=321).detach(), leaf(n_qkv, scale=0.1, seed=322).detach()
    wproj, bproj = leaf(C, C, scale=0.08, seed=323).detach(), leaf(C, scale=0.1, seed=324).detach()
    Ek = Ev = idx = None
    stride, Lk = 0, 0
    if kind != 2:
        rows = 16 if kind == 0 else 128
        Ek, Ev = leaf(rows, KC, scale=0.3, seed=325).detach(), leaf(rows, KC, scale=0.3, seed=326).detach()
        Lk = 16
    if kind == 1:
        stride = 2
        idx, Lk = _msda_idx(T, stride)
    bk, bv = leaf(S, C, scale=0.5, seed=327).detach(), leaf(S, C, scale=0.5, seed=328).detach()
    sa, sp = K.new_site(), K.new_site()
    res = []
    for drain in (False, True):
        F._BRANCH_DRAIN = drain
        try:
            out, o, saved, _trip = F.branch_forward(kind, x, wqkv, bqkv, wproj, bproj, Ek, Ev, bk, bv, idx, stride, Lk,
                                                    attn_drop=(0.1, sa), proj_drop=(0.1, sp), want_o=True, save=True)
            torch.cuda.synchronize()
        finally:
            F._BRANCH_DRAIN = False
        res.append([t.clone() for t in (out, o) + tuple(saved)])
    assert len(res[0]) == len(res[1]) >= 3
    for a_, b_ in zip(*res):
        assert torch.equal(a_, b_)
    assert float(res[0][0].float().abs().max()) > 0


@pytest.mark.parametrize("drop", [0.0, 0.1])
@pytest.mark.parametrize("B,T", [(5, 16), (64, 16), (1030, 16), (3, 64), (130, 64)])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_fused_branch_forward(F, Q, kind, B, T, drop):
    """bf16 fused branch against the fp32 torch chain on the same (bf16-rounded) inputs; with dropout the reference uses the
    exact masks the kernel drew (attention mask from the host RNG replica; proj mask read off the kept entries).
    T = 16, B = 5: a ragged last tile (4 images per workgroup); B = 1030: more tiles than one round of workgroups.
    T = 64 (Tiny-ImageNet's 64 learned tokens / QA-ViT at 32 px): SWA over the four 4x4 windows of the 8x8 grid, MSDA with 40 landmarks
    (three landmark tiles, one key side per image), cross-attention over 64 queries."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    C, H, S, KC = 192, 4, 16, 32
    x = leaf(B, T, C, seed=300).detach().to(torch.bfloat16)
    n_qkv = C if kind == 2 else 3 * C
    wqkv, bqkv = leaf(n_qkv, C, scale=0.08, seed=301).detach(), leaf(n_qkv, scale=0.1, seed=302).detach()
    wproj, bproj = leaf(C, C, scale=0.08, seed=303).detach(), leaf(C, scale=0.1, seed=304).detach()
    Ek = Ev = idx = None
    stride, Lk = 0, 0
    if kind != 2:
        rows = 16 if kind == 0 else 128
        Ek, Ev = leaf(rows, KC, scale=0.3, seed=305).detach(), leaf(rows, KC, scale=0.3, seed=306).detach()
        Lk = 16
    if kind == 1:
        stride = 2
        idx, Lk = _msda_idx(T, stride)
        assert Lk == (10 if T == 16 else 40)
    bk, bv = leaf(1, S, C, scale=0.5, seed=307).detach(), leaf(1, S, C, scale=0.5, seed=308).detach()
    sa, sp = K.new_site(), K.new_site()
    seed, step = [int(v) for v in K.Runtime.get(0).rng.tolist()]
    NK = S if kind == 2 else KC + S
    keep = _attn_keep(seed, step, sa, kind, B, T, H, NK, drop)
    out, o = F.branch_forward(kind, x, wqkv, bqkv, wproj, bproj, Ek, Ev, bk.reshape(S, C), bv.reshape(S, C), idx, stride, Lk,
                              attn_drop=(drop, sa), proj_drop=(drop, sp), want_o=True)
    wq_r, wp_r = wqkv.to(torch.bfloat16).float(), wproj.to(torch.bfloat16).float()         # the kernel reads bf16 weights
    ref, o_ref = _branch_reference(kind, x.float(), wq_r, bqkv, wp_r, bproj, Ek, Ev, bk, bv, idx, stride, keep, drop)
    assert rel(o.reshape(B, T, C), o_ref) <= 3e-2
    if drop > 0:
        kept = out != 0
        frac = kept.float().mean().item()
        assert abs(frac - (1 - drop)) < 0.02, frac
        assert rel(out[kept], (ref / (1 - drop))[kept]) <= 3e-2
        # the proj mask is the GEMM epilogue's contract: drop_factor(key(site), row * C + col)
        from conftest import rng_key, drop_keep
        keep_p = drop_keep(rng_key(seed, step, sp), np.arange(B * T * C, dtype=np.uint64), drop).reshape(B, T, C)
        assert np.array_equal(keep_p, kept.cpu().numpy())
    else:
        assert rel(out, ref) <= 3e-2
    # the projections the kernel saves for the backward pass (k / v through its second, transposed MFMA)
    if drop == 0.0:
        out_s, o_s, saved, _trip = F.branch_forward(kind, x, wqkv, bqkv, wproj, bproj, Ek, Ev, bk.reshape(S, C), bv.reshape(S, C), idx, stride, Lk,
                                                    want_o=True, save=True)
        assert torch.equal(out_s, out) and torch.equal(o_s, o)
        xf32 = x.float().reshape(B * T, C)
        if kind == 0:
            assert rel(saved[0], TF.linear(xf32, wq_r, bqkv)) <= 2e-2
        elif kind == 1:
            pooled_ref = x.float()[:, idx.long()].view(B, -1, stride, C).mean(2).reshape(B * Lk, C)
            assert rel(saved[2], pooled_ref) <= 1e-2
            assert rel(saved[0], TF.linear(xf32, wq_r[:C], bqkv[:C])) <= 2e-2
            assert rel(saved[1], TF.linear(pooled_ref.to(torch.bfloat16).float(), wq_r[C:], bqkv[C:])) <= 2e-2
        else:
            assert rel(saved[0], TF.linear(xf32, wq_r, bqkv)) <= 2e-2
    # the unfused kernels on the same operands agree more tightly (same bf16 rounding points except q/k/v staying in fp32 -> bf16 once)
    if drop == 0.0:
        if kind == 0:
            qkv = F.linear(x, wqkv, bqkv).reshape(B * T, 3 * C)
            nwin = T // 16
            tbl = _win_tbl(T) if nwin > 1 else None
            spec = dict(mode=0, G=B * nwin, Nq=16, L=16, H=H, D=C // H, KC=KC, S=S, groups_per_b=nwin, q_rows_per_b=T, k_rows_per_b=T, q_tbl=tbl, k_tbl=tbl,
                        q_off=0, k_off=C, v_off=2 * C, q_rows=B * T)
            o2 = F.AttnFn.apply(qkv, None, Ek, Ev, bk, bv, spec)
        elif kind == 1:
            pooled = F.GatherPoolFn.apply(x, idx, stride)
            kv = F.linear(pooled, wqkv, bqkv, rows=(C, 2 * C)).reshape(B * Lk, 2 * C)
            q = F.linear(x, wqkv, bqkv, rows=(0, C)).reshape(B * T, C)
            spec = dict(mode=0, G=B, Nq=T, L=Lk, H=H, D=C // H, KC=KC, S=S, groups_per_b=1, q_rows_per_b=T, k_rows_per_b=Lk, q_off=0, k_off=0, v_off=C, q_rows=B * T)
            o2 = F.AttnFn.apply(q, kv, Ek, Ev, bk, bv, spec)
        else:
            q = F.linear(x, wqkv, bqkv).reshape(B * T, C)
            spec = dict(mode=1, G=B, Nq=T, L=0, H=H, D=C // H, S=S, q_off=0, k_off=0, v_off=0, q_rows=B * T)
            o2 = F.AttnFn.apply(q, None, None, None, bk.reshape(S, C), bv.reshape(S, C), spec)
        out2 = F.linear(o2.reshape(B, T, C), wproj, bproj)
        assert rel(out, out2) <= 2e-2


def _proj_keep(seed, step, site, rows, C, p):
    """bool [rows, C] on DEV: the proj-dropout mask of a GEMM epilogue / fused branch kernel (drop_factor(key(site), row * C + col))."""
    from conftest import rng_key, drop_keep
    return torch.from_numpy(drop_keep(rng_key(seed, step, site), np.arange(rows * C, dtype=np.uint64), p).reshape(rows, C)).to(DEV)


@pytest.mark.parametrize("drop", [0.0, 0.1])
@pytest.mark.parametrize("B,T", [(5, 16), (64, 16), (1030, 16), (3, 64), (130, 64)])
@pytest.mark.parametrize("kind", [0, 1, 2])
def test_fused_branch_backward(F, Q, kind, B, T, drop):
    """BranchFn's backward through the fused kernel (proj input gradient + attention-core backward in one launch, csrc/branch_bwd.hip)
    (1) against fp32 torch autograd of the reference chain (HQAViT_CIFAR100.py:441-469 / :496-532 / :613-626) on the same bf16-rounded
    operands with the EXACT dropout masks the kernels drew (attention mask and proj mask from the host RNG replica), every gradient
    compared directly: x, both weights and biases, the Linformer matrices, the shared (bank) rows; and (2) against its backward through
    the unfused kernels (same forward launch, same masks, same saved projections: the two differ only in where bf16 roundings fall)."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    C, H, S, KC = 192, 4, 16, 32
    n_qkv = C if kind == 2 else 3 * C
    idx, stride, Lk = None, 0, 0
    if kind == 1:
        stride = 2
        idx, Lk = _msda_idx(T, stride)
    sa, sp = K.new_site(), K.new_site()
    seed, step = [int(v) for v in K.Runtime.get(0).rng.tolist()]
    gout = leaf(B, T, C, seed=399).detach().to(torch.bfloat16)
    res = []
    for fused in (True, False):
        x = leaf(B, T, C, seed=300).detach().to(torch.bfloat16).requires_grad_(True)
        wqkv, bqkv = leaf(n_qkv, C, scale=0.08, seed=301), leaf(n_qkv, scale=0.1, seed=302)
        wproj, bproj = leaf(C, C, scale=0.08, seed=303), leaf(C, scale=0.1, seed=304)
        Ek = Ev = None
        if kind != 2:
            rows = 16 if kind == 0 else 128
            Ek, Ev = leaf(rows, KC, scale=0.3, seed=305), leaf(rows, KC, scale=0.3, seed=306)
        bk, bv = leaf(1, S, C, scale=0.5, seed=307), leaf(1, S, C, scale=0.5, seed=308)
        if kind == 2:       # cross: the shared rows are activations (projections of the bank), their gradient is returned to autograd
            sk, sv = (bk * 1.0).reshape(S, C), (bv * 1.0).reshape(S, C)
        else:
            sk, sv = bk, bv
        meta = dict(kind=kind, attn_drop=(drop, sa), proj_drop=(drop, sp))
        if kind == 1:
            meta.update(pool_idx=idx, pool_stride=stride, Lk=Lk)
        if kind == 0 and T == 64:
            meta.update(win_tbl=_win_tbl(T))
        old = F._BRANCH_BWD
        F._BRANCH_BWD = fused
        try:
            out = F.BranchFn.apply(x, wqkv, bqkv, wproj, bproj, Ek, Ev, sk, sv, meta)
            out.backward(gout)
        finally:
            F._BRANCH_BWD = old
        torch.cuda.synchronize()
        g = dict(x=x.grad.float(), wqkv=wqkv.grad, bqkv=bqkv.grad, wproj=wproj.grad, bproj=bproj.grad, bk=bk.grad, bv=bv.grad)
        if kind != 2:
            g.update(Ek=Ek.grad, Ev=Ev.grad)
        res.append((out.detach().float(), g))
    (o1, g1), (o2, g2) = res
    assert torch.equal(o1, o2)
    for k_ in g1:
        assert g1[k_] is not None and g2[k_] is not None, k_
        assert torch.isfinite(g1[k_]).all(), k_
        assert rel(g1[k_], g2[k_]) <= 3e-2, (k_, rel(g1[k_], g2[k_]))
    # ---- (1) fp32 torch autograd of the reference chain, exact masks ----
    NK = S if kind == 2 else KC + S
    keep = _attn_keep(seed, step, sa, kind, B, T, H, NK, drop)
    r = dict(x=leaf(B, T, C, seed=300).detach().to(torch.bfloat16).float(),
             wqkv=leaf(n_qkv, C, scale=0.08, seed=301).detach().to(torch.bfloat16).float(), bqkv=leaf(n_qkv, scale=0.1, seed=302).detach(),
             wproj=leaf(C, C, scale=0.08, seed=303).detach().to(torch.bfloat16).float(), bproj=leaf(C, scale=0.1, seed=304).detach(),
             bk=leaf(1, S, C, scale=0.5, seed=307).detach(), bv=leaf(1, S, C, scale=0.5, seed=308).detach())
    if kind != 2:
        rows = 16 if kind == 0 else 128
        r.update(Ek=leaf(rows, KC, scale=0.3, seed=305).detach(), Ev=leaf(rows, KC, scale=0.3, seed=306).detach())
    for v in r.values():
        v.requires_grad_(True)
    ref, _ = _branch_reference(kind, r["x"], r["wqkv"], r["bqkv"], r["wproj"], r["bproj"], r.get("Ek"), r.get("Ev"), r["bk"], r["bv"], idx, stride, keep, drop)
    if drop > 0:
        ref = ref * _proj_keep(seed, step, sp, B * T, C, drop).reshape(B, T, C).float() / (1.0 - drop)
    assert rel(o1, ref) <= 3e-2
    ref.backward(gout.float())
    for k_ in g1:
        assert rel(g1[k_], r[k_].grad) <= 4e-2, (k_, rel(g1[k_], r[k_].grad))


def _cga_reference(x, P, G, H, keep, p_attn):
    """EfficientChannelGroupAttention.forward (HQAViT_CIFAR100.py:559-590, up to ``proj``) as fp32 torch on a name -> tensor dict."""
    B, N, C = x.shape
    cpg = C // G
    xf = x.view(B, N, G, cpg).permute(0, 2, 1, 3).reshape(B * G, N, cpg)
    q = TF.linear(xf, P["q_proj.weight"], P["q_proj.bias"])
    k = TF.linear(xf, P["k_proj.weight"], P["k_proj.bias"])
    v = TF.linear(xf, P["v_proj.weight"], P["v_proj.bias"])
    ccg = q.shape[-1]
    d = ccg // H
    q, k, v = [t.reshape(B * G, N, H, d).transpose(1, 2) for t in (q, k, v)]
    kb = TF.linear(P["bank.global_k"].expand(B, -1, -1), P["bank_k_proj.weight"], P["bank_k_proj.bias"]).unsqueeze(1).expand(-1, G, -1, -1)
    vb = TF.linear(P["bank.global_v"].expand(B, -1, -1), P["bank_v_proj.weight"], P["bank_v_proj.bias"]).unsqueeze(1).expand(-1, G, -1, -1)
    kb = kb.reshape(B * G, -1, H, d).transpose(1, 2)
    vb = vb.reshape(B * G, -1, H, d).transpose(1, 2)
    o = _ref_attn(q, torch.cat([k, kb], 2), torch.cat([v, vb], 2), keep, p_attn)
    o = o.transpose(1, 2).reshape(B * G, N, -1).view(B, G, N, -1).permute(0, 2, 1, 3).reshape(B, N, G * ccg)
    return TF.linear(o, P["proj.weight"], P["proj.bias"])


@pytest.mark.parametrize("drop", [0.0, 0.1])
@pytest.mark.parametrize("B,T", [(3, 16), (64, 16), (1030, 16), (3, 64), (130, 64)])
def test_fused_cga_branch(F, Q, B, T, drop):
    """The channel-group branch through the fused kernel (csrc/cga.hip: q/k/v projections of the six groups, 4 heads of D = 4 over
    tokens + projected bank rows, softmax + dropout, P.V, proj + dropout in one launch) against the unfused chain of the same module:
    identical dropout masks (same sites / counters), so outputs and every gradient agree to bf16 rounding."""
    import importlib
    M = importlib.import_module("qa-vit_amd.modules")
    K = importlib.import_module("qa-vit_amd.kernels")
    cfg = Q.HQAViTConfig()
    cfg.dropout = drop
    res = []
    x0 = leaf(B, T, cfg.embed_dim, seed=600).detach().to(torch.bfloat16)
    g0 = leaf(B, T, cfg.embed_dim, seed=601).detach().to(torch.bfloat16)
    assert F.cga_ok(x0, cfg.num_channel_groups, cfg.num_heads, cfg.global_bank_size)
    for fused in (True, "fwd", False):                         # fused forward + fused backward / fused forward only / unfused chain
        torch.manual_seed(1)
        bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).to(DEV)
        rt = M._Ctx("hqa")
        rt.bank_writes = False
        mod = M.EfficientChannelGroupAttention(cfg, bank, rt).to(DEV).train()
        mod._site, mod._site_attn = 9001, 9002                  # same dropout sites in both runs
        x = x0.clone().requires_grad_(True)
        old = (F._CGA_FUSED, F._CGA_FUSED_BWD)
        F._CGA_FUSED, F._CGA_FUSED_BWD = bool(fused), fused is True
        try:
            out = mod(x)
            out.backward(g0)
        finally:
            F._CGA_FUSED, F._CGA_FUSED_BWD = old
        torch.cuda.synchronize()
        grads = {n: p.grad.clone() for n, p in mod.named_parameters() if p.grad is not None}
        grads.update({"bank." + n: p.grad.clone() for n, p in bank.named_parameters() if p.grad is not None})
        res.append((out.detach().float(), x.grad.float(), grads))
    o2, dx2, g2 = res[2]
    for (o1, dx1, g1) in res[:2]:
        assert rel(o1, o2) <= 3e-2
        if drop > 0:
            assert torch.equal(o1 == 0, o2 == 0)                 # same proj-dropout mask
        assert rel(dx1, dx2) <= 4e-2
        assert set(g1) == set(g2) and len(g1) >= 10
        for k_ in g1:
            assert torch.isfinite(g1[k_]).all(), k_
            assert rel(g1[k_], g2[k_]) <= 4e-2, (k_, rel(g1[k_], g2[k_]))
    # ---- fp32 torch autograd of the reference module (HQAViT_CIFAR100.py:559-595) on the bf16-rounded operands, exact masks ----
    from conftest import attn_keep_mask
    seed, step = [int(v) for v in K.Runtime.get(0).rng.tolist()]
    C, G, H, S = cfg.embed_dim, cfg.num_channel_groups, cfg.num_heads, cfg.global_bank_size
    keep = torch.from_numpy(attn_keep_mask(seed, step, 9002, B * G, H, T, T + S, drop)).to(DEV) if drop > 0 else None
    keep_p = _proj_keep(seed, step, 9001, B * T, C, drop).reshape(B, T, C).float() if drop > 0 else None
    torch.manual_seed(1)
    bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).to(DEV)
    rt = M._Ctx("hqa")
    mod = M.EfficientChannelGroupAttention(cfg, bank, rt).to(DEV)
    P = {n: p.detach().clone() for n, p in mod.named_parameters() if not n.startswith("global_bank.")}
    P.update({"bank." + n: p.detach().clone() for n, p in bank.named_parameters()})
    for n in ("q_proj.weight", "k_proj.weight", "v_proj.weight", "proj.weight"):     # the kernels read bf16 copies of these
        P[n] = P[n].to(torch.bfloat16).float()
    for v in P.values():
        v.requires_grad_(True)
    xr = x0.float().requires_grad_(True)
    ref = _cga_reference(xr, P, G, H, keep, drop)
    if drop > 0:
        ref = ref * keep_p / (1.0 - drop)
    ref.backward(g0.float())
    for (o1, dx1, g1) in res[:2]:
        assert rel(o1, ref) <= 3e-2
        assert rel(dx1, xr.grad) <= 4e-2, rel(dx1, xr.grad)
        for k_ in g1:
            if k_ in P and P[k_].grad is not None:
                assert rel(g1[k_], P[k_].grad) <= 4e-2, (k_, rel(g1[k_], P[k_].grad))
        for k_ in ("q_proj.weight", "k_proj.weight", "v_proj.weight", "proj.weight", "proj.bias", "bank_k_proj.weight", "bank_v_proj.weight",
                   "bank.global_k", "bank.global_v"):
            assert k_ in g1, k_


@pytest.mark.parametrize("B,T", [(5, 16), (1030, 16), (7, 64)])
@pytest.mark.parametrize("kind", [0, 1, 2, "cga"])
def test_fused_branch_nan_rule(F, Q, kind, B, T):
    """efficient_attention's NaN rule (HQAViT_CIFAR100.py:356-357, :394-395) inside the fused branch kernels: one NaN in the input zeroes
    the whole attention output, so every output row is the proj bias (dropout off); the flag words are reset by the launch itself
    (last workgroup), so the next, clean launch is untouched.  B = 1030: more workgroups than one round."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    rt = K.Runtime.get(0)
    assert rt.nan_guard
    C, S, KC = 192, 16, 32
    if kind == "cga" and T != 16 and not F.cga_ok(torch.empty(1, T, C, dtype=torch.bfloat16, device=DEV), 6, 4, S):
        pytest.skip("fused channel-group kernel not built for this token count")
    x = leaf(B, T, C, seed=700).detach().to(torch.bfloat16)
    bad = x.clone()
    bad[B // 2, 7, 11] = float("nan")
    if kind == "cga":
        M = importlib.import_module("qa-vit_amd.modules")
        cfg = Q.HQAViTConfig()
        cfg.dropout = 0.0
        bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).to(DEV)
        ctx = M._Ctx("hqa")
        ctx.bank_writes = False
        mod = M.EfficientChannelGroupAttention(cfg, bank, ctx).to(DEV).eval()
        with torch.no_grad():
            mod.proj.bias.copy_(leaf(C, scale=0.3, seed=701).detach())
        assert F._CGA_FUSED
        run = lambda inp: mod(inp).detach()
        bias = mod.proj.bias.detach()
    else:
        n_qkv = C if kind == 2 else 3 * C
        wqkv, bqkv = leaf(n_qkv, C, scale=0.08, seed=301).detach(), leaf(n_qkv, scale=0.1, seed=302).detach()
        wproj, bproj = leaf(C, C, scale=0.08, seed=303).detach(), leaf(C, scale=0.3, seed=304).detach()
        Ek = Ev = idx = None
        stride, Lk = 0, 0
        if kind != 2:
            rows = 16 if kind == 0 else 128
            Ek, Ev = leaf(rows, KC, scale=0.3, seed=305).detach(), leaf(rows, KC, scale=0.3, seed=306).detach()
            Lk = 16
        if kind == 1:
            stride = 2
            idx, Lk = _msda_idx(T, stride)
        bk, bv = leaf(S, C, scale=0.5, seed=307).detach(), leaf(S, C, scale=0.5, seed=308).detach()
        run = lambda inp: F.branch_forward(kind, inp, wqkv, bqkv, wproj, bproj, Ek, Ev, bk, bv, idx, stride, Lk,
                                           attn_drop=(0.0, 0), proj_drop=(0.0, 0))
        bias = bproj
    clean0 = run(x).clone()
    assert torch.isfinite(clean0).all()
    poisoned = run(bad)
    torch.cuda.synchronize()
    assert rt.nan_flag.tolist() == [0, 0]                    # reset by the launch that used it
    assert torch.equal(poisoned.reshape(-1, C), bias.to(torch.bfloat16).expand(B * T, C))
    clean1 = run(x)
    assert torch.equal(clean1, clean0)


@pytest.mark.parametrize("B,T", [(9, 16), (1030, 16), (5, 64)])
@pytest.mark.parametrize("which", ["swa", "msda", "cga"])
def test_nan_rule_deferred_into_the_bank_write(F, Q, which, B, T):
    """A branch module in training mode writes the bank from its output right after the fused kernel: the NaN rule's rewrite then rides
    in the bank-statistics launch (qavit_bank_stats_nanfix) instead of a launch of its own.  With a poisoned input and dropout ON: the
    output, the bank after the write, the trip word's effect on backward (zero gradient through the branch) and the state left for
    the next, clean call must equal the undeferred path's (QAVIT_DEFER_NANFIX off) bit for bit."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    M = importlib.import_module("qa-vit_amd.modules")
    rt = K.Runtime.get(0)
    C = 192
    cfg = Q.HQAViTConfig() if T == 16 else Q.HQAViTTinyINConfig()
    cfg.dropout = 0.1
    if which == "cga" and not F.cga_ok(torch.empty(1, T, C, dtype=torch.bfloat16, device=DEV), 6, 4, cfg.global_bank_size):
        pytest.skip("fused channel-group kernel not built for this token count")
    x0 = leaf(B, T, C, seed=900).detach().to(torch.bfloat16)
    bad0 = x0.clone()
    bad0[B // 2, 5, 17] = float("nan")
    cls = {"swa": M.EfficientSpatialWindowAttention, "msda": M.EfficientMultiScaleDilatedAttention, "cga": M.EfficientChannelGroupAttention}[which]
    torch.manual_seed(5)
    bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).to(DEV)
    ctx = M._Ctx("hqa")
    mod = cls(cfg, bank, ctx).to(DEV).train()                # ONE module: dropout sites are per module instance
    Q.fill_module(mod)
    state = {k: v.detach().clone() for k, v in mod.state_dict().items()}
    res = {}
    for defer in (True, False):
        F._DEFER_FIX = defer
        mod.load_state_dict(state)
        rt.seed(1234)
        outs = []
        for inp in (bad0, x0):                               # poisoned call, then a clean one through the same state
            xin = inp.clone().requires_grad_(True)
            for p_ in mod.parameters():
                p_.grad = None
            y = mod(xin)
            y.float().square().sum().backward()
            outs += [y.detach().clone(), xin.grad.detach().clone(), bank.global_k.detach().clone(), bank.global_v.detach().clone()]
            rt.advance()
        torch.cuda.synchronize()
        assert rt.nan_flag.tolist() == [0, 0] and rt.pending_fix is None
        res[defer] = outs
    F._DEFER_FIX = True
    assert float(res[True][1].abs().max()) == 0.0            # poisoned call: no gradient reaches x through the branch
    assert torch.isfinite(res[True][4]).all() and float(res[True][5].abs().max()) > 0.0
    for a_, b_ in zip(res[True], res[False]):
        assert torch.equal(a_, b_)


@pytest.mark.parametrize("B,T", [(70, 16), (5, 16), (20, 64)])
def test_qkv_input_gradients_as_one_gemm_in_the_fan_node(F, Q, B, T):
    """FanGroup: the three fused attention branches that read norm1's output (cross q_proj, SWA qkv, MSDA q) leave dq | dk | dv in column
    slices of one [M, 5C] matrix and the fan node's backward runs ONE input-gradient GEMM over the concatenated contraction axis (K = 5C,
    MSDA's landmark-path gradient as the residual addend) instead of three GEMMs whose results it then summed.  A QuadAttentionBlock in
    training mode, dropout on, bf16: output bit-equal (the forward is untouched), input gradient and EVERY parameter gradient against the
    three-GEMM path (QAVIT_DX_CAT off) -- the weight gradients read the same dq bits through a leading dimension of 5C -- and the group
    must really have been used."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    M = importlib.import_module("qa-vit_amd.modules")
    rt = K.Runtime.get(0)
    C = 192
    cfg = Q.HQAViTConfig() if T == 16 else Q.HQAViTTinyINConfig()
    cfg.dropout = 0.1
    x0 = leaf(B, T, C, seed=930).detach().to(torch.bfloat16)
    torch.manual_seed(7)
    bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).to(DEV)
    blk = M.QuadAttentionBlock(cfg, bank, 0.1, M._Ctx("hqa")).to(DEV).train()
    Q.fill_module(blk)
    state = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    gy = torch.randn(B, T, C, device=DEV).to(torch.bfloat16)
    runs = []
    orig = F.FanGroup.run
    F.FanGroup.run = lambda self, *a_, **k_: (runs.append((sorted(self.entries), k_.get("lnbwd") is not None)), orig(self, *a_, **k_))[1]
    res = {}
    try:
        for cat in (True, False):
            F._DX_CAT = cat
            blk.load_state_dict(state)
            rt.seed(2468)
            del runs[:]
            for p_ in blk.parameters():
                p_.grad = None
            xin = x0.clone().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y = blk(xin)
            y.backward(gy)
            torch.cuda.synchronize()
            res[cat] = dict(y=y.detach().float().clone(), dx=xin.grad.float().clone(),
                            **{n_: p_.grad.detach().clone() for n_, p_ in blk.named_parameters() if p_.grad is not None})
            # one group run holding cross (2), SWA (0) and MSDA (1), with norm1's backward as the GEMM's epilogue where the K-loop kernel
            # takes the shape (M >= 1024 rows) -- or none
            assert runs == ([([0, 1, 2], B * T >= 1024)] if cat else []), runs
    finally:
        F.FanGroup.run = orig
        F._DX_CAT = True
    assert torch.equal(res[True]["y"], res[False]["y"])
    assert set(res[True]) == set(res[False])
    for k_ in res[True]:
        if k_ == "y":
            continue
        a_, b_ = res[True][k_], res[False][k_]
        # one rounding of the summed input gradient instead of three roundings and a sum: bf16-level differences downstream of norm1
        assert rel(a_, b_) <= (2e-2 if k_ == "dx" or k_.startswith("norm1") else 1e-5 if ".qkv." in k_ or "cross_attn.q_proj" in k_ else 2e-2), (k_, rel(a_, b_))


def test_a_failed_call_does_not_leave_a_deferred_nan_rule_behind(F, Q):
    """The deferred NaN rule is per-device state between the fused branch launch and the bank write that consumes it.  A call that fails
    in between (here: the bank write raises) must drop it, or every later forward would raise 'never consumed'."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    M = importlib.import_module("qa-vit_amd.modules")
    rt = K.Runtime.get(0)
    cfg = Q.HQAViTConfig()
    bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).to(DEV)
    mod = M.EfficientSpatialWindowAttention(cfg, bank, M._Ctx("hqa")).to(DEV).train()
    Q.fill_module(mod)
    x = leaf(4, 16, 192, seed=77).detach().to(torch.bfloat16)
    orig = F.bank_write
    seen = []

    def boom(*a, **k):
        seen.append(rt.pending_fix is not None)
        raise RuntimeError("injected failure between the deferral and its consumer")
    F.bank_write = boom
    try:
        with pytest.raises(RuntimeError, match="injected"):
            mod(x)
    finally:
        F.bank_write = orig
    assert seen == [True]                                   # the rule WAS deferred when the failure hit
    assert rt.pending_fix is None
    y = mod(x)                                              # and the next call runs
    torch.cuda.synchronize()
    assert torch.isfinite(y.float()).all() and rt.pending_fix is None and rt.nan_flag.tolist() == [0, 0]


@pytest.mark.parametrize("B,T", [(7, 16), (3, 64)])
def test_nan_rule_of_the_cross_branch_rides_in_the_compress_fuse_launch(F, Q, B, T):
    """Inside a QuadAttentionBlock the cross branch's output is read next by the compress-fuse launch, which then carries the branch's NaN
    rule (qavit_cfuse_args.fix) instead of a launch of its own.  A block in training mode, dropout ON, one poisoned input (all four
    rules trip), then a clean one through the same state: output, input gradient, every parameter gradient and the bank must equal the
    path with the rule's own launch (QAVIT_DEFER_NANFIX_CFUSE off) bit for bit, and the deferral must really have happened."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    M = importlib.import_module("qa-vit_amd.modules")
    rt = K.Runtime.get(0)
    C = 192
    cfg = Q.HQAViTConfig() if T == 16 else Q.HQAViTTinyINConfig()
    cfg.dropout = 0.1
    x0 = leaf(B, T, C, seed=910).detach().to(torch.bfloat16)
    bad0 = x0.clone()
    bad0[B // 2, 3, 11] = float("nan")
    torch.manual_seed(6)
    bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).to(DEV)
    ctx = M._Ctx("hqa")
    blk = M.QuadAttentionBlock(cfg, bank, 0.1, ctx).to(DEV).train()
    Q.fill_module(blk)
    state = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    calls = []
    orig = F._defer_fix
    F._defer_fix = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    res, ndef = {}, {}
    try:
        for defer in (True, False):
            F.DEFER_FIX_CFUSE = defer
            blk.load_state_dict(state)
            rt.seed(4321)
            outs = []
            del calls[:]
            for inp in (bad0, x0):
                xin = inp.clone().requires_grad_(True)
                for p_ in blk.parameters():
                    p_.grad = None
                with torch.autocast("cuda", dtype=torch.bfloat16):
                    y = blk(xin)
                y.float().square().sum().backward()
                outs += [("y", y.detach().clone()), ("dx", xin.grad.detach().clone()), ("bank", bank.global_k.detach().clone())]
                outs += [(n_, p_.grad.detach().clone()) for n_, p_ in sorted(blk.named_parameters()) if p_.grad is not None]
                rt.advance()
            torch.cuda.synchronize()
            assert rt.nan_flag.tolist() == [0, 0] and rt.pending_fix is None
            res[defer], ndef[defer] = outs, len(calls)
    finally:
        F._defer_fix = orig
        F.DEFER_FIX_CFUSE = True
    assert ndef[True] == ndef[False] + 2, ndef               # one more deferral per block call: the cross branch's
    assert len(res[True]) == len(res[False])
    # Bit equality where the arithmetic ORDER is fixed: activations, the input gradient, the bank, and the narrow parameter gradients that
    # leave as partial rows folded in a fixed order (the layer scale ccf_ffn.gamma, the fusion logits).  Every other parameter gradient of
    # this un-armed backward is summed with float atomics from several workgroups (LayerNorm / CCF partials, weight-gradient tile flushes):
    # their last bits depend on arrival order, run to run, so they are compared to summation-order tolerance (round 3 asserted bit equality
    # on all of them and failed on a 1-ulp difference in ccf_ffn.gamma, then still five atomics on one address).
    fixed = ("y", "dx", "bank", "ccf_ffn.gamma", "fusion.fusion_weights")
    for (na, a_), (nb_, b_) in zip(res[True], res[False]):
        assert na == nb_
        a_, b_ = torch.nan_to_num(a_.float()), torch.nan_to_num(b_.float())
        if na in fixed:
            assert torch.equal(a_, b_), na
        else:
            assert float((a_ - b_).abs().max()) <= 2e-6 * float(b_.abs().max()) + 1e-30, na
    # (the block's residual carries the poisoned input row through the first call: only equality is asserted there)
    n1 = len(res[True]) // 2
    assert all(torch.isfinite(t.float()).all() for _, t in res[True][n1:])       # the clean call after it is clean


def test_narrow_parameter_gradients_are_bit_reproducible(F, Q):
    """Scalar layer scales and two-logit blends are summed over the whole tensor.  One float atomic per workgroup on ONE address gave a
    last-bit difference from run to run (the red test of round 3: ccf_ffn.gamma, 265.3792 vs 265.3793).  They now leave as narrow partial
    rows folded in a fixed order by qavit_ln_param_reduce: the same launch repeated gives the same bits, at grids of hundreds of workgroups,
    with the reduce launched at once (eager) -- and the sum matches fp32 torch."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    rt = K.Runtime.get(0)
    rows, C = 16384, 192                                    # 384 workgroups in the vector kernels
    for dtype in (torch.bfloat16, torch.float32):
        dy = torch.randn(rows, C, device=DEV).to(dtype)
        u = torch.randn(rows, C, device=DEV).to(dtype)
        a = torch.randn(rows, C, device=DEV).to(dtype)
        gam = torch.full((1,), 0.3, device=DEV)
        fw = torch.tensor([0.4, -0.3], device=DEV)
        seen = {"scale_add": [], "mix2": [], "mix3": []}
        for _ in range(4):
            dg = torch.zeros(1, device=DEV)
            K.scale_add_bwd(dy, u, gam, torch.empty_like(dy), dg, rows, C, (0.1, 77, 16), rt.rng)
            seen["scale_add"].append(dg.clone())
            d2 = torch.zeros(2, device=DEV)
            K.mix2_bwd(dy, a, u, fw, torch.empty_like(dy), torch.empty_like(dy), d2)
            seen["mix2"].append(d2.clone())
            d3 = torch.zeros(2, device=DEV)
            K.mix3_bwd(dy, a, u, u, fw, torch.empty_like(dy), torch.empty_like(dy), torch.empty_like(dy), d3, (0.0, 5), rt.rng)
            seen["mix3"].append(d3.clone())
        torch.cuda.synchronize()
        for name, vals in seen.items():
            assert all(torch.equal(vals[0], v) for v in vals[1:]), (name, [v.tolist() for v in vals])
            assert float(vals[0].abs().max()) > 0, name
        # value check (drop-path off for the reference): dgamma = sum dy * u
        dg = torch.zeros(1, device=DEV)
        K.scale_add_bwd(dy, u, gam, torch.empty_like(dy), dg, rows, C, (0.0, 0, 16), rt.rng)
        ref = (dy.double() * u.double()).sum()
        assert abs(float(dg) - float(ref)) <= 1e-3 * math.sqrt(rows * C), (float(dg), float(ref))      # |sum| ~ sqrt(n); fp32 partial sums
        w = torch.softmax(fw.double(), 0)
        ds = torch.stack([(dy.double() * a.double()).sum(), (dy.double() * u.double()).sum()])
        ref2 = w * (ds - (ds * w).sum())
        assert rel(seen["mix2"][0], ref2.float()) <= 1e-3


@pytest.mark.parametrize("B,T", [(6, 16), (3, 64)])
@pytest.mark.parametrize("kind", [0, 1, 2, "cga"])
def test_fused_branch_nan_rule_backward(F, Q, kind, B, T):
    """The NaN rule in the fused BACKWARD kernels: efficient_attention returns zeros_like(q) when it sees a NaN (HQAViT_CIFAR100.py:356-357,
    :394-395), a tensor without history -- so after a tripped forward the gradient through the branch is exactly zero for x, the q/k/v
    weights, the Linformer matrices and the bank rows; dW_proj = dz^T 0 = 0; only db_proj = colsum(dout) flows (proj(0) = bias).
    The next, clean call through the same kernels has ordinary gradients."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    M = importlib.import_module("qa-vit_amd.modules")
    rt = K.Runtime.get(0)
    assert rt.nan_guard
    C, S, KC = 192, 16, 32
    gout = leaf(B, T, C, seed=811).detach().to(torch.bfloat16)
    if kind == "cga":
        if not F.cga_ok(torch.empty(1, T, C, dtype=torch.bfloat16, device=DEV), 6, 4, S):
            pytest.skip("fused channel-group kernel not built for this token count")
        cfg = Q.HQAViTConfig()
        cfg.dropout = 0.0
        bank = M.GlobalTokenBank(cfg.global_bank_size, cfg.embed_dim).to(DEV)
        ctx = M._Ctx("hqa")
        ctx.bank_writes = False
        mod = M.EfficientChannelGroupAttention(cfg, bank, ctx).to(DEV).train()
        params = dict(mod.named_parameters())
        params = {n: p for n, p in params.items() if not n.startswith("norm.") and ("global_bank." not in n or n.endswith(("global_k", "global_v")))}
        run = lambda inp: mod(inp)
        pbias, pw = "proj.bias", "proj.weight"
    else:
        n_qkv = C if kind == 2 else 3 * C
        params = dict(wqkv=leaf(n_qkv, C, scale=0.08, seed=301), bqkv=leaf(n_qkv, scale=0.1, seed=302), wproj=leaf(C, C, scale=0.08, seed=303), bproj=leaf(C, scale=0.3, seed=304),
                      bk=leaf(1, S, C, scale=0.5, seed=307), bv=leaf(1, S, C, scale=0.5, seed=308))
        Ek = Ev = idx = None
        stride, Lk = 0, 0
        if kind != 2:
            rows = 16 if kind == 0 else 128
            params.update(Ek=leaf(rows, KC, scale=0.3, seed=305), Ev=leaf(rows, KC, scale=0.3, seed=306))
        meta = dict(kind=kind, attn_drop=(0.0, 0), proj_drop=(0.0, 0))
        if kind == 1:
            stride = 2
            idx, Lk = _msda_idx(T, stride)
            meta.update(pool_idx=idx, pool_stride=stride, Lk=Lk)
        if kind == 0 and T == 64:
            meta.update(win_tbl=_win_tbl(T))

        def run(inp):
            sk, sv = params["bk"], params["bv"]
            if kind == 2:
                sk, sv = (sk * 1.0).reshape(S, C), (sv * 1.0).reshape(S, C)
            return F.BranchFn.apply(inp, params["wqkv"], params["bqkv"], params["wproj"], params["bproj"], params.get("Ek"), params.get("Ev"), sk, sv, meta)
        pbias, pw = "bproj", "wproj"
    x0 = leaf(B, T, C, seed=810).detach().to(torch.bfloat16)
    clean_bk = None if kind == "cga" else params["bk"].detach().clone()
    for poisoned in (True, False):
        x = x0.clone()
        # the NaN enters through the shared key rows (k_full has a NaN; x stays clean so that "zero" is checkable for the q/k/v weights:
        # their gradient GEMM multiplies the zero dq/dk/dv with x).  Channel-group branch: through x itself (its shared rows are
        # projections of the bank computed upstream); there the q/k/v weight gradients are 0 . NaN and are not checked.
        if kind == "cga":
            if poisoned:
                x[B // 2, 5, 17] = float("nan")
        else:
            with torch.no_grad():
                params["bk"].copy_(clean_bk)
                if poisoned:
                    params["bk"][0, 3, 40] = float("nan")
        x.requires_grad_(True)
        for p_ in params.values():
            p_.grad = None
        out = run(x)
        out.backward(gout)
        torch.cuda.synchronize()
        assert rt.nan_flag.tolist() == [0, 0]
        if poisoned:
            assert torch.equal(out.detach().reshape(-1, C), params[pbias].detach().to(torch.bfloat16).expand(B * T, C))
            assert float(x.grad.float().abs().max()) == 0.0
            for n, p_ in params.items():
                if n == pbias:
                    assert rel(p_.grad, gout.float().reshape(-1, C).sum(0)) <= 1e-2
                elif kind == "cga" and n in ("q_proj.weight", "k_proj.weight", "v_proj.weight"):
                    continue
                else:
                    assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, n
        else:
            assert torch.isfinite(x.grad.float()).all() and float(x.grad.float().abs().max()) > 0
            for n, p_ in params.items():
                assert p_.grad is not None and torch.isfinite(p_.grad).all() and float(p_.grad.abs().max()) > 0, n


@pytest.mark.parametrize("B,T", [(3, 16), (64, 16), (1030, 16), (3, 64), (130, 64)])
def test_fused_compress_fuse(F, Q, B, T):
    """CompressFuseFn through the one-launch forward (csrc/cfuse.hip: four LayerNorms, four Linear(192 -> 48), concat, softmax-weighted
    scaling) against the three-launch forward, and both against torch; the backward (shared) runs on what each forward saved.
    T = 64: the node is token-wise, an image's 64 tokens go to the kernel as four 16-token problems."""
    C, Cb, nb = 192, 48, 4
    outs = []
    for fused in (True, "fwd", False):                         # fused forward + backward / fused forward only / unfused
        xs = [leaf(B, T, C, seed=700 + i).detach().to(torch.bfloat16).requires_grad_(True) for i in range(nb)]
        gs = [leaf(C, scale=0.2, seed=710 + i) for i in range(nb)]
        bs = [leaf(C, scale=0.2, seed=720 + i) for i in range(nb)]
        Ws = [leaf(Cb, C, scale=0.08, seed=730 + i) for i in range(nb)]
        bi = [leaf(Cb, scale=0.1, seed=740 + i) for i in range(nb)]
        with torch.no_grad():
            for g_ in gs:
                g_.add_(1.0)
        fw = leaf(nb, scale=0.5, seed=750)
        args = []
        for i in range(nb):
            args += [xs[i], gs[i], bs[i], Ws[i], bi[i]]
        old = (F._CFUSE, F._CFUSE_BWD)
        F._CFUSE, F._CFUSE_BWD = bool(fused), fused is True
        try:
            y = F.CompressFuseFn.apply(fw, 1e-5, *args)
            go = leaf(B, T, nb * Cb, seed=760).detach().to(torch.bfloat16)
            y.backward(go)
        finally:
            F._CFUSE, F._CFUSE_BWD = old
        torch.cuda.synchronize()
        outs.append((y.detach().float(), [x.grad.float() for x in xs], [t.grad.clone() for t in gs + bs + Ws + bi + [fw]]))
        if fused is True:                                    # torch reference on the same (bf16-rounded) inputs
            w = torch.softmax(fw.detach(), 0)
            ref = torch.cat([TF.linear(TF.layer_norm(xs[i].detach().float(), (C,), gs[i].detach(), bs[i].detach()).to(torch.bfloat16).float(),
                                       Ws[i].detach().to(torch.bfloat16).float(), bi[i].detach()) * w[i] for i in range(nb)], -1)
            assert rel(y, ref) <= 3e-2
            # ... and fp32 torch autograd of HybridFusion(concat_i compress_i(norm_i(.))) (HQAViT_CIFAR100.py:1075-1081) for EVERY gradient
            xr = [x.detach().float().requires_grad_(True) for x in xs]
            pr = [t.detach().clone().requires_grad_(True) for t in gs + bs] + [t.detach().to(torch.bfloat16).float().requires_grad_(True) for t in Ws] + \
                 [t.detach().clone().requires_grad_(True) for t in bi + [fw]]
            gr, br_, Wr, bir, fwr = pr[:nb], pr[nb:2 * nb], pr[2 * nb:3 * nb], pr[3 * nb:4 * nb], pr[4 * nb]
            wr_ = torch.softmax(fwr, 0)
            ref2 = torch.cat([TF.linear(TF.layer_norm(xr[i], (C,), gr[i], br_[i]), Wr[i], bir[i]) * wr_[i] for i in range(nb)], -1)
            ref2.backward(go.float())
            ref_grads = ([t.grad for t in xr], [t.grad for t in pr])
    y2, dx2, g2 = outs[2]
    for (y1, dx1, g1) in outs[:2]:
        assert rel(y1, y2) <= 1e-2
        for a_, b_ in zip(dx1 + g1, dx2 + g2):
            assert rel(a_, b_) <= 3e-2
    names = [f"x{i}" for i in range(nb)] + [f"{n}{i}" for n in ("gamma", "beta", "W", "bias") for i in range(nb)] + ["fusion_weights"]
    for (y1, dx1, g1) in outs:                               # all three paths against fp32 torch autograd
        for nm, a_, b_ in zip(names, dx1 + g1, ref_grads[0] + ref_grads[1]):
            assert rel(a_, b_) <= 4e-2, (nm, rel(a_, b_))


@pytest.mark.parametrize("drop,dp", [(0.0, 0.0), (0.1, 0.2)])
@pytest.mark.parametrize("M,T", [(16 * 5 + 0, 16), (1000, 8), (16384, 16)])
def test_fused_mlp2(F, Q, M, T, drop, dp):
    """BottleneckMLP + residual in one launch each way (csrc/mlp2.hip) against (1) fp32 torch autograd of
    x + drop_path(dropout(fc2(dropout(gelu(fc1(y)))))) (HQAViT_CIFAR100.py:651-656, :1082-1083) on the bf16-rounded operands with the
    exact masks (host RNG replica), every gradient compared, and (2) the two-GEMM chain it replaces (same masks by contract).
    M = 1000: a ragged last tile of 64 rows."""
    import importlib
    from conftest import rng_key, drop_keep
    K = importlib.import_module("qa-vit_amd.kernels")
    C, Hd = 192, 96
    s1, s2, sp = K.new_site(), K.new_site(), K.new_site()
    seed, step = [int(v) for v in K.Runtime.get(0).rng.tolist()]
    gout = leaf(M, C, seed=905).detach().to(torch.bfloat16)
    res = []
    for fused in (True, False):
        y = leaf(M, C, seed=900).detach().to(torch.bfloat16).requires_grad_(True)
        x = leaf(M, C, seed=901).detach().to(torch.bfloat16).requires_grad_(True)
        w1, b1 = leaf(Hd, C, scale=0.08, seed=902), leaf(Hd, scale=0.1, seed=903)
        w2, b2 = leaf(C, Hd, scale=0.1, seed=904), leaf(C, scale=0.1, seed=906)
        if fused:
            assert F.mlp2_ok(y, x, w1, w2)
            out = F.Mlp2Fn.apply(y, x, w1, b1, w2, b2, dict(drop1=(drop, s1), drop2=(drop, s2), dp=(dp, sp, T)))
        else:
            h = F.linear(y, w1, b1, act="gelu", drop=(drop, s1))
            out = F.linear(h, w2, b2, drop=(drop, s2), dp=(dp, sp, T), resid=x)
        out.backward(gout)
        torch.cuda.synchronize()
        res.append((out.detach().float(), dict(y=y.grad.float(), x=x.grad.float(), w1=w1.grad, b1=b1.grad, w2=w2.grad, b2=b2.grad)))
    (o1, g1), (o2, g2) = res
    assert rel(o1, o2) <= 1e-2
    for k_ in g1:
        assert rel(g1[k_], g2[k_]) <= 3e-2, (k_, rel(g1[k_], g2[k_]))
    # fp32 torch autograd with the exact masks
    r = dict(y=leaf(M, C, seed=900).detach().to(torch.bfloat16).float(), x=leaf(M, C, seed=901).detach().to(torch.bfloat16).float(),
             w1=leaf(Hd, C, scale=0.08, seed=902).detach().to(torch.bfloat16).float(), b1=leaf(Hd, scale=0.1, seed=903).detach(),
             w2=leaf(C, Hd, scale=0.1, seed=904).detach().to(torch.bfloat16).float(), b2=leaf(C, scale=0.1, seed=906).detach())
    for v in r.values():
        v.requires_grad_(True)
    h = TF.gelu(TF.linear(r["y"], r["w1"], r["b1"]))
    if drop > 0:
        h = h * torch.from_numpy(drop_keep(rng_key(seed, step, s1), np.arange(M * Hd, dtype=np.uint64), drop).reshape(M, Hd)).to(DEV).float() / (1 - drop)
    u = TF.linear(h, r["w2"], r["b2"])
    if drop > 0:
        u = u * torch.from_numpy(drop_keep(rng_key(seed, step, s2), np.arange(M * C, dtype=np.uint64), drop).reshape(M, C)).to(DEV).float() / (1 - drop)
    if dp > 0:
        keep = torch.from_numpy(drop_keep(rng_key(seed, step, sp), np.arange((M + T - 1) // T, dtype=np.uint64), dp)).to(DEV).float() / (1 - dp)
        u = u * keep.repeat_interleave(T)[:M, None]
    ref = r["x"] + u
    assert rel(o1, ref) <= 2e-2
    ref.backward(gout.float())
    for k_ in g1:
        assert rel(g1[k_], r[k_].grad) <= 4e-2, (k_, rel(g1[k_], r[k_].grad))


def test_partial_row_reduce(F, Q):
    """qavit_ln_param_reduce: dst halves += column sums of n partial rows, dense rows and rows embedded in a wider record (stride) --
    the end-of-backward fold of the LayerNorm dgamma/dbeta partials and of the fused branch backward's dE / shared-row partials."""
    import importlib
    K = importlib.import_module("qa-vit_amd.kernels")
    g = torch.Generator().manual_seed(7)
    for n, C, stride, off in ((256, 192, 0, 0), (3, 64, 0, 0), (258, 512, 7168, 0), (77, 1536, 7168, 1024), (40, 1536, 7168, 4096)):
        width = stride if stride else 2 * C
        parts = torch.randn(n, width, generator=g).to(DEV)
        dg = torch.randn(C, generator=g).to(DEV)
        db = torch.randn(C, generator=g).to(DEV)
        ref_g = dg + parts[:, off:off + C].sum(0)
        ref_b = db + parts[:, off + C:off + 2 * C].sum(0)
        K.reduce_now([K.DeferredLN.desc(parts.data_ptr() + off * 4, n, C, dg.data_ptr(), db.data_ptr(), stride)])
        torch.cuda.synchronize()
        assert rel(dg, ref_g) <= 1e-5 and rel(db, ref_b) <= 1e-5, (n, C, stride)
    # one half only
    parts = torch.randn(9, 384, generator=g).to(DEV)
    dg = torch.zeros(192, device=DEV)
    K.reduce_now([K.DeferredLN.desc(parts.data_ptr(), 9, 192, dg.data_ptr(), None, 0)])
    assert rel(dg, parts[:, :192].sum(0)) <= 1e-5


def test_bank_proj2(F, Q):
    """The bank projections of the cross / channel-group branches as one grouped skinny launch each way against two F.linear calls:
    values, weight / bias gradients (which must see the SNAPSHOT of the bank, not its later in-place update) and the bank rows' own
    gradient, accumulated in place by the GEMM's residual epilogue on top of what .grad already holds."""
    S, C, n = 16, 192, 192
    res = []
    for fused in (True, False):
        gk, gv = leaf(1, S, C, scale=0.5, seed=500), leaf(1, S, C, scale=0.5, seed=501)
        lk, lv = torch.nn.Linear(C, n).to(DEV), torch.nn.Linear(C, n).to(DEV)
        with torch.no_grad():
            for i, lin in enumerate((lk, lv)):
                lin.weight.copy_(leaf(n, C, scale=0.1, seed=502 + i).detach())
                lin.bias.copy_(leaf(n, scale=0.1, seed=504 + i).detach())
        gk.grad = torch.full_like(gk, 0.25)                  # something to accumulate onto
        bank = type("B", (), {})()
        bank.global_k, bank.global_v = gk, gv
        snap = (gk.detach().clone(), gv.detach().clone())
        if fused:
            yk, yv = F.bank_proj2(bank, snap, lk, lv)
        else:
            a, b = F.bank_snapshot(bank, snap)
            yk = F.linear(a, lk.weight, lk.bias).reshape(S, n)
            yv = F.linear(b, lv.weight, lv.bias).reshape(S, n)
        with torch.no_grad():                                # the in-place bank write that follows in the model
            gk.mul_(3.0)
        go = leaf(S, n, seed=506).detach()
        (yk * go).sum().backward(retain_graph=True)
        (yv * go * 0.5).sum().backward()
        torch.cuda.synchronize()
        res.append(dict(yk=yk.detach(), yv=yv.detach(), gk=gk.grad.clone(), gv=gv.grad.clone(), wk=lk.weight.grad.clone(), bk=lk.bias.grad.clone(),
                        wv=lv.weight.grad.clone(), bv=lv.bias.grad.clone()))
    for k_ in res[0]:
        assert rel(res[0][k_], res[1][k_]) <= 2e-5, k_
    x0 = leaf(1, S, C, scale=0.5, seed=500).detach().reshape(S, C)
    assert rel(res[0]["wk"], leaf(S, n, seed=506).detach().t() @ x0) <= 2e-5      # forward-time bank, not 3x


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,C", [(4, 100), (37, 10), (1024, 100), (1500, 200)])
def test_cross_entropy_label_smoothing(F, dtype, B, C):
    """One-kernel CE (loss + gradient) against torch's nn.CrossEntropyLoss(label_smoothing), plain and as the MixUp pair loss."""
    lg = leaf(B, C, seed=400).detach().to(dtype).requires_grad_(True)
    g = torch.Generator().manual_seed(5)
    ya, yb = torch.randint(0, C, (B,), generator=g).to(DEV), torch.randint(0, C, (B,), generator=g).to(DEV)
    lam = torch.tensor([0.37], device=DEV)
    for ls in (0.0, 0.12):
        for mixed in (False, True):
            lg.grad = None
            loss = F.cross_entropy(lg, ya, ls, y_b=yb if mixed else None, lam=lam if mixed else None)
            (loss * 3.0).backward()
            r = lg.detach().float().clone().requires_grad_(True)
            ref = TF.cross_entropy(r, ya, label_smoothing=ls)
            if mixed:
                ref = 0.37 * ref + 0.63 * TF.cross_entropy(r, yb, label_smoothing=ls)
            (ref * 3.0).backward()
            assert abs(float(loss) - float(ref)) <= 2e-5 * abs(float(ref)), (ls, mixed)
            assert rel(lg.grad, r.grad) <= (1e-5 if dtype == torch.float32 else 1e-2), (ls, mixed)


def test_integration_example_runs(Q):
    """INTEGRATION.md's ctypes listing (tools/integration_example.py) runs as written and matches SDPA."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "integration_example.py")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "max-rel error" in r.stdout
