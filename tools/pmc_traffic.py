"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of tools/prof_step.py) into HBM bytes per
kernel launch for the entry-point families bench.py reports -> profiles/traffic.json.

gfx950 corrections (MI355X_MICROARCH.md, HBM): both counters are in KiB; FETCH_SIZE tallies 128-byte requests of a
wide coalesced read at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte-per-lane stores and float atomics.
usage: pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, json, sys
FAM = [("branch_fwd", ("branch_fwd_kernel",)), ("branch_bwd", ("branch_bwd_kernel",)), ("cga_fwd", ("cga_fwd_kernel", "cga64_fwd_kernel")), ("cga_bwd", ("cga_bwd_kernel", "cga64_bwd_kernel")), ("mlp2_fwd", ("mlp2_fwd_kernel",)), ("mlp2_bwd", ("mlp2_bwd_kernel",)), ("cfuse_fwd", ("cfuse_fwd_kernel",)), ("cfuse_bwd", ("cfuse_bwd_kernel",)), ("gemm_nt", ("gemm_nt_",)), ("gemm_tn_grouped", ("gemm_tn_",)),
       ("attn_bwd", ("true>(qavit_attn_args", "attn_bwd_kernel", "attn_reduce")), ("attn_fwd", ("false>(qavit_attn_args", "attn_fwd_kernel")),
       ("layernorm_bwd", ("layernorm_bwd",)), ("layernorm_fwd", ("layernorm_fwd",)), ("row_stats", ("row_stats",)),
       ("dwconv_bwd", ("dwconv_bwd",)), ("dwconv_fwd", ("dwconv_fwd",)), ("ccf_mid_bwd", ("ccf_bwd",)), ("ccf_mid_fwd", ("ccf_fwd",)),
       ("bank_stats", ("bank_stats", "bank_reduce")), ("bn", ("bn_",)), ("tl_fwd", ("tl_fwd_kernel",)), ("tl_bwd", ("tl_bwd_kernel",))]
def fold(path):
    agg = {}
    for r in csv.DictReader(open(path)):
        nm = r["Kernel_Name"]
        fam = next((f for f, keys in FAM if any(k in nm for k in keys)), None)
        if fam is None:
            continue
        d = agg.setdefault(fam, [0, 0.0])
        d[0] += 1; d[1] += float(r["Counter_Value"])
    return agg
f, w = fold(sys.argv[1]), fold(sys.argv[2])
out = {}
for fam in f:
    n = f[fam][0]
    rd = 2.0 * f[fam][1] * 1024.0 / n
    wr = w.get(fam, [n, 0.0])[1] * 1024.0 / max(w.get(fam, [n, 0.0])[0], 1)
    out[fam] = {"hbm_bytes_per_kernel": round(rd + wr), "read_bytes": round(rd), "write_bytes": round(wr), "kernels_profiled": n,
                "note": "FETCH_SIZE x2 (gfx950) + WRITE_SIZE, KiB -> bytes; eager step, B=1024 bf16"}
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_kernel"] * kv[1]["kernels_profiled"]):
    print(f"{k:16s} n={v['kernels_profiled']:5d}  {v['hbm_bytes_per_kernel']/1e6:9.2f} MB/kernel  (read {v['read_bytes']/1e6:8.2f}  write {v['write_bytes']/1e6:8.2f})")
