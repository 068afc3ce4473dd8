"""Fold a rocprofv3 kernel_stats.csv into the C-ABI entry-point families bench.py's roofline reports.
usage: kfamily.py <kernel_stats.csv> <steps-profiled>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
# every training step ends with ONE fused AdamW launch: its call count is the number of steps the trace holds (warm-up, capture and
# instrumented eager steps included), a better divisor than the caller's guess
_adamw = [int(r["Calls"]) for r in rows if "adamw_kernel" in r["Name"]]
if _adamw:
    n = float(sum(_adamw))
FAM = [("branch_fwd", ("branch_fwd_kernel", "branch_nan_fix")), ("branch_bwd", ("branch_bwd_kernel",)), ("cga (fused)", ("cga_fwd_kernel", "cga_bwd_kernel", "cga64_fwd_kernel", "cga64_bwd_kernel")), ("compress-fuse (fused)", ("cfuse_fwd_kernel", "cfuse_bwd_kernel")), ("mlp2 (fused)", ("mlp2_fwd_kernel", "mlp2_bwd_kernel")), ("gemm_nt", ("gemm_nt_",)), ("gemm_tn", ("gemm_tn_", "tnu_table_write")), ("attn_bwd", ("attn2_kernel<", "attn3_kernel<", "attn4_kernel<", "attn_bwd_kernel", "attn_reduce")),
       ("layernorm_bwd", ("layernorm_bwd", "mix3_ln_bwd")), ("layernorm_fwd", ("layernorm_fwd", "mix3_ln_fwd")), ("row_stats", ("row_stats",)), ("dwconv", ("dwconv",)),
       ("ccf", ("ccf_",)), ("bank", ("bank_",)), ("bn", ("bn_",)), ("tokmix/upmix/tl", ("tokmix", "upmix", "tl_fwd_kernel", "tl_bwd_kernel")), ("elementwise (own)", ("hybrid_", "scale_add", "mix2_", "mix3_", "dropout_kernel", "gather_pool", "token_mean", "patchify", "im2col", "col2im", "nan_", "zero_f32", "pack_kernel", "adamw", "l2_", "rng_", "ce_ls_kernel", "sum_k_kernel", "copy2_kernel", "gate_mix", "local_clip", "chan_scale", "sln_", "ln_param_reduce", "ln_dadd")), ("bench harness", ("spin_kernel",)), ("torch/other", ("",))]
agg = {}
for r in rows:
    nm = r["Name"]
    fam = next(f for f, keys in FAM if any(k in nm for k in keys))
    if fam == "attn_bwd" and ("false>" in nm or "attn_fwd_kernel" in nm):
        fam = "attn_fwd"
    if fam == "attn_bwd" and "true>" not in nm and "attn_reduce" not in nm and "attn_bwd_kernel" not in nm:
        fam = "attn_fwd"
    d = agg.setdefault(fam, [0, 0])
    d[0] += int(r["Calls"]); d[1] += int(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
print(f"{'family':16s} {'kernels/step':>12s} {'ms/step':>9s} {'avg us':>8s} {'share':>6s}")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:16s} {v[0]/n:12.1f} {v[1]/1e6/n:9.3f} {v[1]/1e3/max(v[0],1):8.2f} {100*v[1]/tot:5.1f}%")
print(f"{'total':16s} {sum(v[0] for v in agg.values())/n:12.1f} {tot/1e6/n:9.3f}")
