"""Does a memset node misbehave under back-to-back hipGraph replays on this stack, independently of this package?
A graph of N x [hipMemsetAsync(buf, 0) -> kernel that atomically adds a known total into buf -> copy the total out]
-- the exact structure of torch's `sum` semaphore zeroing -- is replayed R times without host synchronisation and every
replay's result is checked afterwards (each replay appends its value to a log on the device).
usage: python tools/graph_memset_probe.py [replays]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from importlib import import_module
gi = import_module("qa-vit_amd.graphinfo")
R = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = "cuda"
x = torch.ones(16 << 20, device=dev)
acc = torch.zeros(64, device=dev)
log = torch.zeros(R, 16, device=dev)
idx = torch.zeros(1, dtype=torch.int64, device=dev)
row = torch.zeros(16, device=dev)
def body():
    for k in range(16):
        acc.zero_()                                # -> memset node
        acc[:1].add_(x[k << 20:(k + 1) << 20].sum())       # a multi-block torch reduction: hipMemsetAsync of its semaphore buffer + the kernel
        row[k:k + 1].copy_(acc[:1])
    log.index_copy_(0, idx, row.unsqueeze(0))
    idx.add_(1)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    body(); idx.zero_()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph(keep_graph=True)
with torch.cuda.graph(g):
    body()
print("node kinds:", gi.node_kinds(g.raw_cuda_graph()))
for _ in range(R):
    g.replay()
torch.cuda.synchronize()
bad = (log != float(1 << 20)).nonzero()
print(f"replays {R}, wrong entries {bad.shape[0]} of {log.numel()}", bad[:8].tolist(), log[bad[:8, 0], bad[:8, 1]].tolist() if bad.numel() else "")
