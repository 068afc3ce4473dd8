"""Is a replayed training step bound by the host's hipGraphLaunch or by the device?  Times the replay() CALL on the CPU (no sync) against
the device time per step, for the default workload.  If the call takes about as long as the step, the host enqueues nodes in lockstep
with the device and every graph node costs host time whether or not its kernel is short."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=True), total_steps=100000, warmup_steps=10, compute_dtype=torch.bfloat16)
tr.capture(x, y)
for _ in range(20):
    tr.replay()
torch.cuda.synchronize()
# (a) back-to-back replays: host time per call while the device queue is full
t0 = time.perf_counter(); calls = []
for _ in range(30):
    a = time.perf_counter(); tr.replay(); calls.append(time.perf_counter() - a)
t_enq = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
# (b) one replay on an idle device: host time of the call alone, then the wait
torch.cuda.synchronize()
a = time.perf_counter(); tr.replay(); c1 = time.perf_counter() - a
torch.cuda.synchronize(); c2 = time.perf_counter() - a
calls.sort()
print(f"30 back-to-back replays: host time in replay() {t_enq * 1e3 / 30:.2f} ms per call (median {calls[15] * 1e3:.2f}, min {calls[0] * 1e3:.2f}); wall per step {t_all * 1e3 / 30:.2f} ms")
print(f"one replay on an idle device: call returns after {c1 * 1e3:.2f} ms, step complete after {c2 * 1e3:.2f} ms")
print("graph nodes:", getattr(tr, "graph_nodes", None))
