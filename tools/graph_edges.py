"""Dump the captured training step's hipGraph (hipGraphDebugDotPrint) and print, for chosen kernels, which nodes they depend on: is a
chain that should be independent (token path vs CNN lateral path) really independent in the graph the executor sees?
usage: graph_edges.py [kernel-name-substring ...]"""
import sys, os, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import qavit_amd as Q
B = 256
model = Q.HQAViT(Q.HQAViTConfig()); Q.fill_module(model); model = model.cuda().train()
g = torch.Generator().manual_seed(1234)
x = torch.randn(B, 3, 32, 32, generator=g).cuda(); y = torch.randint(0, 100, (B,), generator=g).cuda()
tr = Q.Trainer(model, Q.TrainingConfig(use_amp=True), total_steps=100000, warmup_steps=10, compute_dtype=torch.bfloat16)
path = "gpurun_out/step_graph.dot"
os.makedirs("gpurun_out", exist_ok=True)
tr.capture(x, y, debug_dot=path)
print("census:", tr.graph_nodes)
txt = open(path).read()
labels = dict(re.findall(r'"?(\w+)"?\s*\[[^\]]*label="([^"]*)"', txt))
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
print(len(labels), "labelled nodes,", len(edges), "edges")
preds = {}
for a, b in edges:
    preds.setdefault(b, []).append(a)
succs = {}
for a, b in edges:
    succs.setdefault(a, []).append(b)
multi = [n for n, p in preds.items() if len(p) > 1]
fan = [n for n, p in succs.items() if len(p) > 1]
print("nodes with more than one predecessor:", len(multi), " nodes with more than one successor:", len(fan))
for n in fan[:12]:
    print("FORK", labels.get(n, n)[:70], "->", [labels.get(s, s)[:40] for s in succs[n]])
for n in multi[:12]:
    print("JOIN", labels.get(n, n)[:70], "<-", [labels.get(s, s)[:40] for s in preds[n]])
for key in sys.argv[1:]:
    for n, lab in labels.items():
        if key in lab:
            print(key, ":", lab[:80], "<-", [labels.get(p_, p_)[:60] for p_ in preds.get(n, [])])
            break
