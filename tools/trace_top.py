"""Per-(kernel, grid) totals of a rocprofv3 kernel-trace CSV, for the LAST `n` steps of tools/prof_step.py."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
d = collections.OrderedDict()
for r in rows:
    name = r.get("Kernel_Name", "")
    key = (name.replace("void ", "")[:58], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("Grid_Size_Z"))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    d.setdefault(key, []).append(dur)
items = sorted(d.items(), key=lambda kv: -sum(kv[1]))
tot = sum(sum(v) for v in d.values())
print(f"total {tot/1e3/nsteps:.2f} ms/step")
for k, v in items[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print(f"{sum(v)/1e3/nsteps:7.3f} ms/step  n={len(v)//nsteps:4d}  avg={sum(v)/len(v):8.1f}us  grid=({k[1]},{k[2]},{k[3]})  {k[0]}")
