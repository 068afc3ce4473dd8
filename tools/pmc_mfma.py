"""Fold one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAVE_CYCLES
GRBM_GUI_ACTIVE; program = tools/prof_step.py) into MFMA utilisation per kernel family -> profiles/mfma_util.json.

MfmaUtil as rocprofv3's own derived counter defines it (rocprofv3 -L): sum(SQ_VALU_MFMA_BUSY_CYCLES) / (GRBM_GUI_ACTIVE
* SIMD_NUM); the per-dispatch GRBM_GUI_ACTIVE in the CSV is the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS note), so
it is divided by 8; SIMD_NUM = 256 CUs x 4.  MFMA FLOPs executed = SQ_INSTS_VALU_MFMA_MOPS_BF16 * 512 (operands padded
to the 16x16x16 tile included -- this is what the matrix pipe did, not the algorithmic count).
usage: pmc_mfma.py <counter_collection.csv> <out.json> [steps-profiled]"""
import csv, json, sys
FAM = [("attn_fused", ("branch_fwd_kernel",)), ("attn_fused_bwd", ("branch_bwd_kernel",)), ("cga_fused", ("cga_fwd_kernel", "cga_bwd_kernel", "cga64_fwd_kernel", "cga64_bwd_kernel")), ("mlp2_fused", ("mlp2_fwd_kernel", "mlp2_bwd_kernel")), ("cfuse_fused", ("cfuse_fwd_kernel", "cfuse_bwd_kernel")),
       ("gemm_nt", ("gemm_nt_",)), ("gemm_tn", ("gemm_tn_",)),
       ("attn_bwd", ("true>(qavit_attn_args", "attn_bwd_kernel")), ("attn_reduce", ("attn_reduce",)),
       ("attn_fwd", ("false>(qavit_attn_args", "attn_fwd_kernel")),
       ("layernorm", ("layernorm_", "row_stats")), ("dwconv", ("dwconv",)), ("ccf", ("ccf_",)), ("bank", ("bank_",)),
       ("bn", ("bn_",)), ("tokmix/upmix/tl", ("tokmix", "upmix", "tl_fwd_kernel", "tl_bwd_kernel")), ("other", ("",))]
SIMD_NUM, XCDS = 1024, 8
steps = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
agg = {}
disp = {}
rows_in = list(csv.DictReader(open(sys.argv[1])))
# keep the LAST `steps` training steps only (model construction / weight filling / warm-up kernels precede them):
# a step ends with the fused AdamW kernel
ends = sorted({int(r["Dispatch_Id"]) for r in rows_in if "adamw" in r["Kernel_Name"]})
if len(ends) > steps:
    lo, hi = ends[-int(steps) - 1], ends[-1]
    rows_in = [r for r in rows_in if lo < int(r["Dispatch_Id"]) <= hi]
for r in rows_in:
    nm = r["Kernel_Name"]
    fam = next(f for f, keys in FAM if any(k in nm for k in keys))
    d = agg.setdefault(fam, {})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    disp.setdefault(fam, set()).add(r["Dispatch_Id"])
out = {}
tot = {}
for fam, c in agg.items():
    for k, v in c.items():
        tot[k] = tot.get(k, 0.0) + v
rows = list(agg.items()) + [("whole step", tot)]
for fam, c in rows:
    gui = c.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
    busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
    out[fam] = {
        "kernels_per_step": (len(disp[fam]) if fam in disp else sum(len(v) for v in disp.values())) / steps,
        "mfma_util_pct": round(100.0 * busy / max(gui * SIMD_NUM, 1.0), 3),
        "mfma_busy_cycles": busy, "gui_active_cycles": gui,
        "mfma_gflop_executed_per_step": round(c.get("SQ_INSTS_VALU_MFMA_MOPS_BF16", 0.0) * 512 / 1e9 / steps, 3),
        "sq_busy_cycles": c.get("SQ_BUSY_CYCLES", 0.0), "sq_wave_cycles": c.get("SQ_WAVE_CYCLES", 0.0),
    }
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(f"{'family':14s} {'kern/step':>9s} {'MfmaUtil%':>9s} {'GFLOP(mfma)/step':>17s} {'share of active':>15s}")
for fam, v in sorted(out.items(), key=lambda kv: -kv[1]["gui_active_cycles"]):
    print(f"{fam:14s} {v['kernels_per_step']:9.1f} {v['mfma_util_pct']:9.2f} {v['mfma_gflop_executed_per_step']:17.2f} {100*v['gui_active_cycles']/max(out['whole step']['gui_active_cycles'],1):14.1f}%")
