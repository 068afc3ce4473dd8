import re, collections, sys
txt = open("gpurun_out/step_graph.dot").read()
labels = re.findall(r'\[[^\]]*label="([^"]*)"', txt)
c = collections.Counter()
for l in labels:
    l2 = re.sub(r'\\n.*', '', l)
    c[l2[:110]] += 1
for k, v in c.most_common():
    if 'qv' in k or 'qavit' in k: continue
    print(v, k)
print("total labelled", len(labels))
