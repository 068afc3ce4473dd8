"""Do kernels of a rocprofv3 --kernel-trace overlap?  For the steady-state part of the trace (last 60 %): wall time, sum of kernel
durations, overlap = sum / wall, the time at least two kernels run concurrently, and the share of kernel time spent in launches whose
grid is smaller than the 256 CUs (workgroups < 256).  usage: overlap_report.py <kernel_trace.csv>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0),
              int(r.get("Workgroup_Size", r.get("Workgroup_Size_X", 1)) or 1)) for r in rows), key=lambda e: e[0])
ev = ev[int(len(ev) * 0.4):]
t0, t1 = ev[0][0], max(e[1] for e in ev)
wall = t1 - t0
busy = sum(e[1] - e[0] for e in ev)
pts = sorted([(e[0], 1) for e in ev] + [(e[1], -1) for e in ev])
conc, last, two, any_ = 0, t0, 0, 0
for t, d in pts:
    if conc >= 2: two += t - last
    if conc >= 1: any_ += t - last
    conc += d; last = t
small = sum(e[1] - e[0] for e in ev if e[3] // max(e[4], 1) < 256)
print(f"kernels {len(ev)}  wall {wall/1e6:.2f} ms  sum of durations {busy/1e6:.2f} ms  (sum/wall = {busy/wall:.3f})")
print(f"time with >= 1 kernel running {any_/1e6:.2f} ms ({100*any_/wall:.1f} % of wall), >= 2 concurrently {two/1e6:.2f} ms ({100*two/wall:.1f} %)")
print(f"kernel time in launches of < 256 workgroups: {small/1e6:.2f} ms ({100*small/busy:.1f} % of kernel time)")
