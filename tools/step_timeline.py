"""One replayed training step as a timeline: every kernel of the LAST step in a rocprofv3 kernel trace (k_kernel_trace.csv) in start
order with its duration, the idle gap before it on its queue and whether another kernel was running beside it.
usage: step_timeline.py <k_kernel_trace.csv> [first] [count]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")) for r in rows]
ks.sort()
ends = [i for i, k in enumerate(ks) if "adamw_kernel" in k[2]]
lo, hi = ends[-2] + 1, ends[-1] + 1
step = ks[lo:hi]
t0 = step[0][0]
print(f"{len(step)} kernels, {(step[-1][1] - t0) / 1e3:.1f} us from first start to last end; busy (union) ", end="")
busy, cur_s, cur_e = 0, step[0][0], step[0][1]
for s, e, _, _ in step[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print(f"{busy / 1e3:.1f} us, sum of durations {sum(e - s for s, e, _, _ in step) / 1e3:.1f} us")
first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
count = int(sys.argv[3]) if len(sys.argv) > 3 else len(step)
prev_end = {}
run_end = 0
for i, (s, e, n, q) in enumerate(step):
    gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
    overl = s < run_end
    if first <= i < first + count:
        short = n.replace("void ", "").replace("qv::(anonymous namespace)::", "").replace("qv::", "")[:70]
        print(f"{i:4d} t={(s - t0) / 1e3:9.1f} dur={(e - s) / 1e3:7.1f} gap={gap:6.1f} q={q[-3:]:>3s} {'||' if overl else '  '} {short}")
    prev_end[q] = e
    run_end = max(run_end, e)
