"""Device time of the TokenLearner mix and the TokenUpMix kernels on their own (forward, backward), replayed from a hipGraph.

usage: python3 tools/bench_tokens.py [B=1024] [N=64] [M=16] [reps=30]        (N patch tokens, M learned tokens; C = 192)
Environment knobs of the kernels under test apply (QAVIT_UPMIX_BWD_GRID ...)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import qavit_amd as Q  # noqa: E402
from importlib import import_module  # noqa: E402

F = import_module("qa-vit_amd.functional")
K = import_module("qa-vit_amd.kernels")
PARTS = os.environ.get("BENCH_TOKENS_PARTS", "1") != "0"      # parameter gradients as partial rows + one reduce launch, as inside a training step


def timed(fn, reps):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        with torch.cuda.graph(g, stream=s):
            for _ in range(4):
                fn()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps / 4 * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    M = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    reps = int(sys.argv[4]) if len(sys.argv) > 4 else 30
    Q.lib.load()
    dev, dt, C = "cuda", torch.bfloat16, 192
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, N, C, generator=g).to(dev).to(dt).requires_grad_(True)
    scores = torch.randn(B, N, M, generator=g).to(dev).to(dt).requires_grad_(True)
    xc = torch.randn(B, M, C, generator=g).to(dev).to(dt).requires_grad_(True)
    W = (torch.randn(N, M, generator=g) * 0.2).to(dev).requires_grad_(True)
    bias = torch.zeros(N, device=dev, requires_grad=True)
    gam = torch.ones(C, device=dev, requires_grad=True)
    bet = torch.zeros(C, device=dev, requires_grad=True)
    for p in (W, bias, gam, bet):
        p.grad = torch.zeros_like(p)
    gy = torch.randn(B, N, C, generator=g).to(dev).to(dt)
    gxc = torch.randn(B, M, C, generator=g).to(dev).to(dt)

    t_mix_f = timed(lambda: F.TokMixFn.apply(scores.detach(), x.detach()), reps)
    t_up_f = timed(lambda: F.UpMixFn.apply(xc.detach(), W.detach(), bias.detach(), gam.detach(), bet.detach(), 1e-5), reps)

    def mix_fb():
        scores.grad = None; x.grad = None
        F.TokMixFn.apply(scores, x).backward(gxc)

    def up_fb():
        xc.grad = None
        K.DeferredLN.enabled = PARTS
        try:
            F.UpMixFn.apply(xc, W, bias, gam, bet, 1e-5).backward(gy)
            K.DeferredLN.flush()
        finally:
            K.DeferredLN.enabled = False

    t_mix_fb = timed(mix_fb, reps)
    t_up_fb = timed(up_fb, reps)
    mb = lambda *n: sum(n) * 2 / 1e6          # noqa: E731
    print(f"B={B} N={N} M={M}:  tokmix fwd {t_mix_f:7.1f} us   fwd+bwd {t_mix_fb:7.1f} us (bwd ~{t_mix_fb - t_mix_f:6.1f};"
          f" algorithmic bytes fwd {mb(B * N * C, B * N * M, B * M * C):.1f} MB, bwd {mb(2 * B * N * C, 2 * B * N * M, 2 * B * M * C):.1f} MB)")
    print(f"                   upmix  fwd {t_up_f:7.1f} us   fwd+bwd {t_up_fb:7.1f} us (bwd ~{t_up_fb - t_up_f:6.1f};"
          f" algorithmic bytes fwd {mb(B * M * C, B * N * C):.1f} MB, bwd {mb(B * N * C, 2 * B * M * C):.1f} MB)")


if __name__ == "__main__":
    main()
