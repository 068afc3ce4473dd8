"""Fold the MFMA counter pass over tools/bench_branch.py (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16
SQ_WAVE_CYCLES GRBM_GUI_ACTIVE) per kernel variant: MfmaUtil = sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs), rocprofv3's own
derived-counter formula.  Also prints the utilisation with the EMPTY kernel's GUI-active cycles subtracted: branch_nan_fix_kernel does nothing when
no NaN was seen, and what it reads per launch under counter collection is the profiler's own serialisation cost, paid by every kernel of the pass.
usage: pmc_branch.py <counter_collection.csv>"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for r in rows:
    k = r["Kernel_Name"]
    m = re.search(r"(branch_\w+<[^>]*>|branch_nan_fix_kernel|cga\w*_kernel|cfuse_\w+_kernel|mlp2_\w+_kernel(?:<\w+>)?|gemm_nt_big_kernel<[^>]*>|attn3_kernel<[^>]*>)", k)
    if not m:
        continue
    k = m.group(1)
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
empty = agg["branch_nan_fix_kernel"]["GRBM_GUI_ACTIVE"] / 8 / max(len(n["branch_nan_fix_kernel"]), 1) if "branch_nan_fix_kernel" in agg else 0.0
print(f"{'kernel':46s} {'calls':>5s} {'MfmaUtil%':>9s} {'gui cyc/launch':>14s} {'busy cyc/SIMD':>13s} {'MFMA GFLOP/launch':>17s} {'util% net of the empty-kernel cycles':>38s}")
for k, c in sorted(agg.items()):
    d = len(n[k]); gui = c["GRBM_GUI_ACTIVE"] / 8
    busy = c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024
    net = 100 * busy / d / max(gui / d - empty, 1.0) if k != "branch_nan_fix_kernel" else 0.0
    print(f"{k:46s} {d:5d} {100 * busy / max(gui, 1):9.2f} {gui / d:14.0f} {busy / d:13.0f} {c['SQ_INSTS_VALU_MFMA_MOPS_BF16'] * 512 / 1e9 / d:17.2f} {net:38.2f}")
print(f"\nempty kernel (branch_nan_fix_kernel, no NaN seen): {empty:.0f} GUI-active cycles per launch under counter collection")
