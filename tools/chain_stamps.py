"""Where the two chains of the captured training step run, WITHOUT a profiler attached.

QAVIT_STAMPS=1 makes the model take wall-clock stamps (one-lane kernels, qavit_stamp) at the fork / join points of the step: the
start and end of the CNN lateral chain and of the token path in forward, their mirrors in backward, the weight-gradient flush and
the optimiser.  The stamps are graph nodes like any other kernel, so a replayed hipGraph records them at full speed; rocprofv3's
kernel trace slows the host's graph launch enough that the step becomes host-paced (12.2 ms against 11.05 ms) and its timeline
shows fork delays the free-running step may not have.  This tool replays the step back to back and prints the stamps of the LAST replay
relative to its first stamp, then the same for a replay that starts on an idle device.

usage: QAVIT_STAMPS=1 python3 tools/chain_stamps.py [batch=1024] [replays=30]
"""
import os
import sys
import time

os.environ["QAVIT_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import qavit_amd as Q  # noqa: E402
from importlib import import_module  # noqa: E402

K = import_module("qa-vit_amd.kernels")
par = import_module("qa-vit_amd.parallel")


def show(title, ms):
    st = K.Stamps.read()
    print(f"== {title}: {ms:.3f} ms per step")
    for n, t in sorted(st.items(), key=lambda kv: kv[1]):
        print(f"  {t:9.1f} us  {n}")
    g = lambda a: st.get(a)          # noqa: E731
    if g("lat.begin") is not None and g("tok.begin") is not None:
        print(f"  forward: token path starts {g('tok.begin') - g('lat.begin'):+.1f} us after the lateral chain; lateral chain done at "
              f"{g('lat.R4.f'):.1f}, token stage 1 done at {g('tok.stage1.f'):.1f}")
        print(f"  backward: token path reaches the patch embedding at {g('tok.embed.b'):.1f}, lateral chain reaches its first conv at "
              f"{g('lat.stem0.b'):.1f}, weight-gradient flush queued at {g('dw.flush'):.1f}, backward joined at {g('bwd.end'):.1f}, "
              f"step ends {g('step.end'):.1f}")


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    Q.lib.load()
    dev = torch.device("cuda", 0)
    cfg = Q.HQAViTConfig()
    model = Q.HQAViT(cfg)
    Q.fill_module(model)
    model = model.to(dev).train()
    g = torch.Generator().manual_seed(1234)
    x = torch.randn(B, 3, cfg.img_size, cfg.img_size, generator=g).to(dev)
    y = torch.randint(0, cfg.num_classes, (B,), generator=g).to(dev)
    tr = Q.Trainer(model, Q.TrainingConfig(batch_size=B, use_amp=True), total_steps=100000, warmup_steps=1000,
                   compute_dtype=torch.bfloat16, order=par.bucket_order)
    tr.capture(x, y, with_optim=True, warmup=3)
    for _ in range(10):
        tr.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        tr.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    show(f"last of {n} back-to-back replays (lateral stream {'on' if os.environ.get('QAVIT_LATERAL_STREAM', '1') != '0' else 'off'}, "
         f"order {os.environ.get('QAVIT_LATERAL_ORDER', '0')})", ms)
    time.sleep(0.05)
    t0 = time.perf_counter()
    tr.replay()
    torch.cuda.synchronize()
    show("one replay on an idle device", (time.perf_counter() - t0) * 1e3)


if __name__ == "__main__":
    main()
