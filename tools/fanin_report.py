"""Where does autograd still sum gradients itself?  Walks the autograd graph of one forward pass and lists every (node, output) that
more than one consumer feeds a gradient into -- each is an elementwise add kernel the engine issues in backward (the model's own
fan-outs go through FanOutFn / alias outputs, whose sums happen inside a kernel that runs anyway)."""
import sys, os, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
model.compute_dtype = torch.bfloat16
x = torch.randn(B, 3, 32, 32).cuda()
out = model(x)
loss = out.float().sum()
indeg = collections.Counter()
seen = set()
stack = [loss.grad_fn]
while stack:
    fn = stack.pop()
    if fn is None or fn in seen:
        continue
    seen.add(fn)
    for nxt, idx in fn.next_functions:
        if nxt is None:
            continue
        indeg[(nxt, idx)] += 1
        stack.append(nxt)
rows = collections.Counter()
for (fn, idx), n in indeg.items():
    if n > 1 and type(fn).__name__ != "AccumulateGrad":
        rows[(type(fn).__name__, idx, n)] += 1
for (name, idx, n), c in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f"x{c:3d}  {name} output {idx}: {n} consumers -> {n - 1} add(s) each")
acc = sum(1 for (fn, idx), n in indeg.items() if n > 1 and type(fn).__name__ == "AccumulateGrad")
print("leaf parameters that receive more than one autograd gradient:", acc)
