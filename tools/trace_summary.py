"""Group a rocprofv3 kernel-trace CSV by (kernel, grid) and print mean durations."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.OrderedDict()
for r in rows:
    name = r.get("Kernel_Name", "")
    if pat and pat not in name: continue
    key = (name[:60], r.get("Grid_Size_X"), r.get("Grid_Size_Y"), r.get("LDS_Block_Size"), r.get("VGPR_Count"), r.get("SGPR_Count"))
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    d.setdefault(key, []).append(dur)
for k, v in d.items():
    v2 = sorted(v)
    print(f"{k[0]:60s} grid=({k[1]},{k[2]}) lds={k[3]} vgpr={k[4]} sgpr={k[5]} n={len(v):3d} med={v2[len(v2)//2]:8.1f}us min={v2[0]:8.1f}us")
