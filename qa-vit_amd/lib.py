"""ctypes binding of libqavit_hip.so (include/qavit.h).  No fallback: if the library is missing or a call
fails, a RuntimeError is raised -- the product path never computes on the CPU."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("QAVIT_LIB") or os.path.join(_HERE, "libqavit_hip.so")     # QAVIT_LIB: A/B another build of the same C-ABI

F32, BF16 = 0, 1

i32, i64, f32, vp = C.c_int, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("M", i32), ("N", i32), ("K", i32),
        ("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp), ("ldc", i64), ("bias", vp),
        ("a_mode", i32), ("ln_gamma", vp), ("ln_beta", vp), ("ln_eps", f32), ("ln_mean", vp), ("ln_rstd", vp),
        ("a_Z", vp), ("a_ldz", i64), ("a_act", i32), ("a_drop_p", f32), ("a_drop_site", i32),
        ("a_dp_p", f32), ("a_dp_site", i32), ("a_dp_rows", i32), ("a_scale", f32), ("a_out", vp), ("a_ldo", i64),
        ("Z", vp), ("ldz", i64), ("act", i32), ("drop_p", f32), ("drop_site", i32), ("scale", f32),
        ("dp_p", f32), ("dp_site", i32), ("dp_rows", i32), ("R", vp), ("ldr", i64), ("rng", vp),
        ("A2", vp), ("lda2", i64), ("a2_k0", i32),
        ("e_x", vp), ("e_mean", vp), ("e_rstd", vp), ("e_gamma", vp), ("e_add0", vp), ("e_add1", vp),
        ("e_dgamma", vp), ("e_dbeta", vp), ("e_parts", vp),
    ]


class LnReduceDesc(C.Structure):
    _fields_ = [("parts", vp), ("nparts", i32), ("C", i32), ("dgamma", vp), ("dbeta", vp), ("stride", i64)]


class NanFix(C.Structure):                         # struct qavit_nan_fix
    _fields_ = [("flag", vp), ("trip", vp), ("bias", vp), ("drop_p", f32), ("drop_site", i32), ("rng", vp),
                ("o_save", vp), ("ldos", i64), ("Co", i32)]


class CgaArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("B", i32), ("T", i32), ("C", i32), ("G", i32), ("H", i32), ("D", i32), ("S", i32),
        ("x", vp), ("ldx", i64), ("wqkv_rm", vp), ("bqkv", vp), ("wproj_rm", vp), ("bproj", vp), ("sh_k", vp), ("sh_v", vp),
        ("out", vp), ("ldo", i64), ("o_save", vp),
        ("attn_drop_p", f32), ("attn_drop_site", i32), ("proj_drop_p", f32), ("proj_drop_site", i32), ("rng", vp), ("nan_flag", vp), ("nan_trip", vp), ("nan_defer", i32),
    ]


class CfuseArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("B", i32), ("T", i32), ("C", i32), ("NB", i32), ("CB", i32),
        ("x", vp * 4), ("gamma", vp * 4), ("beta", vp * 4), ("w_rm", vp * 4), ("bias", vp * 4),
        ("fw", vp), ("eps", f32), ("cat", vp), ("y", vp), ("mean", vp * 4), ("rstd", vp * 4),
        ("fix", NanFix), ("fix_branch", i32),
    ]


class CfuseBwdArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("B", i32), ("T", i32), ("C", i32), ("NB", i32), ("CB", i32),
        ("dy", vp), ("cat", vp), ("x", vp * 4), ("gamma", vp * 4), ("w_rm", vp * 4), ("mean", vp * 4), ("rstd", vp * 4),
        ("fw", vp), ("dcat", vp), ("dx", vp * 4), ("parts", vp),
    ]


class CgaBwdArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("B", i32), ("T", i32), ("C", i32), ("G", i32), ("H", i32), ("D", i32), ("S", i32),
        ("dout", vp), ("lddout", i64), ("x", vp), ("ldx", i64), ("wqkv_rm", vp), ("wqkvT_rm", vp), ("bqkv", vp), ("wprojT_rm", vp),
        ("sh_k", vp), ("sh_v", vp),
        ("attn_drop_p", f32), ("attn_drop_site", i32), ("proj_drop_p", f32), ("proj_drop_site", i32), ("rng", vp),
        ("dz", vp), ("lddz", i64), ("dqkv", vp), ("dx", vp), ("lddx", i64), ("parts", vp), ("nan_trip", vp),
    ]


class Mlp2Args(C.Structure):
    _fields_ = [
        ("dtype", i32), ("M", i32), ("C", i32), ("Hd", i32),
        ("y", vp), ("ldy", i64), ("resid", vp), ("ldr", i64), ("w1_rm", vp), ("b1", vp), ("w2_rm", vp), ("b2", vp),
        ("drop1_p", f32), ("drop1_site", i32), ("drop2_p", f32), ("drop2_site", i32), ("dp_p", f32), ("dp_site", i32), ("dp_rows", i32), ("rng", vp),
        ("out", vp), ("ldo", i64), ("z1", vp), ("h1", vp),
    ]


class Mlp2BwdArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("M", i32), ("C", i32), ("Hd", i32),
        ("g", vp), ("ldg", i64), ("z1", vp), ("w1_rm", vp), ("w2_rm", vp),
        ("drop1_p", f32), ("drop1_site", i32), ("drop2_p", f32), ("drop2_site", i32), ("dp_p", f32), ("dp_site", i32), ("dp_rows", i32), ("rng", vp),
        ("dz2", vp), ("dz1", vp), ("dy", vp), ("lddy", i64),
    ]


class BranchBwdArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("kind", i32), ("B", i32), ("T", i32), ("C", i32), ("H", i32), ("D", i32), ("KC", i32), ("S", i32), ("L", i32),
        ("dout", vp), ("lddout", i64), ("wprojT_frag", vp), ("q", vp), ("ldq", i64),
        ("k_tok", vp), ("v_tok", vp), ("ldkv", i64), ("kv_rows", i32), ("o", vp), ("ldo", i64),
        ("E_k", vp), ("E_v", vp), ("sh_k", vp), ("sh_v", vp),
        ("attn_drop_p", f32), ("attn_drop_site", i32), ("proj_drop_p", f32), ("proj_drop_site", i32), ("rng", vp),
        ("dz", vp), ("lddz", i64), ("dq", vp), ("lddq", i64), ("dk_tok", vp), ("dv_tok", vp), ("lddkv", i64),
        ("parts", vp), ("parts_stride", i64), ("nan_trip", vp),
    ]


class GemmTnArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("M", i32), ("N", i32), ("K", i32),
        ("A", vp), ("lda", i64), ("B", vp), ("ldb", i64), ("C", vp), ("ldc", i64), ("colsum", vp),
        ("ln_gamma", vp), ("ln_beta", vp), ("ln_mean", vp), ("ln_rstd", vp), ("splits", i32),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("mode", i32),
        ("G", i32), ("Nq", i32), ("L", i32), ("H", i32), ("D", i32), ("KC", i32), ("S", i32),
        ("groups_per_b", i32), ("q_rows_per_b", i32), ("k_rows_per_b", i32), ("q_tbl", vp), ("k_tbl", vp),
        ("q", vp), ("ldq", i64), ("k_tok", vp), ("ldk", i64), ("v_tok", vp), ("ldv", i64),
        ("E_k", vp), ("E_v", vp), ("sh_k", vp), ("sh_v", vp), ("o", vp), ("ldo", i64), ("nan_flag", vp),
        ("d_o", vp), ("lddo", i64), ("dq", vp), ("lddq", i64), ("dk_tok", vp), ("lddk", i64), ("dv_tok", vp), ("lddv", i64),
        ("ws", vp), ("ws_floats", i64), ("dE_k", vp), ("dE_v", vp), ("dsh_k", vp), ("dsh_v", vp),
        ("drop_p", f32), ("drop_site", i32), ("rng", vp),
    ]


class TlArgs(C.Structure):                          # struct qavit_tl_args
    _fields_ = [("x", vp), ("ln_g", vp), ("ln_b", vp), ("eps", f32), ("W", vp), ("bias", vp),
                ("p", vp), ("xc", vp), ("mean", vp), ("rstd", vp), ("B", i32), ("N", i32), ("M", i32), ("C", i32)]


class TlBwdArgs(C.Structure):                       # struct qavit_tl_bwd_args
    _fields_ = [("dxc", vp), ("x", vp), ("p", vp), ("mean", vp), ("rstd", vp), ("ln_g", vp), ("ln_b", vp), ("W", vp),
                ("dx", vp), ("parts", vp), ("B", i32), ("N", i32), ("M", i32), ("C", i32)]


class BranchArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("kind", i32),
        ("B", i32), ("T", i32), ("C", i32), ("H", i32), ("D", i32), ("KC", i32), ("S", i32), ("L", i32),
        ("x", vp), ("ldx", i64), ("wqkv_frag", vp), ("bqkv", vp), ("wproj_frag", vp), ("bproj", vp),
        ("E_k", vp), ("E_v", vp), ("sh_k", vp), ("sh_v", vp), ("pool_idx", vp), ("pool_stride", i32),
        ("out", vp), ("ldo", i64), ("o_save", vp),
        ("attn_drop_p", f32), ("attn_drop_site", i32), ("proj_drop_p", f32), ("proj_drop_site", i32), ("rng", vp),
        ("nan_flag", vp), ("drain_waits", i32),
        ("q_save", vp), ("ldq_save", i64), ("kv_save", vp), ("ldkv_save", i64), ("pooled_save", vp), ("nan_trip", vp), ("nan_defer", i32),
    ]


class CcfArgs(C.Structure):
    _fields_ = [
        ("dtype", i32), ("flags", i32), ("B", i32), ("Hs", i32), ("Ws", i32), ("C", i32),
        ("h", vp), ("out", vp), ("g1", vp), ("b1", vp), ("g2", vp), ("b2", vp), ("eps", f32),
        ("w", vp), ("cbias", vp), ("cscale", vp), ("mean1", vp), ("rstd1", vp), ("mean2", vp), ("rstd2", vp),
        ("d_out", vp), ("d_h", vp), ("dg1", vp), ("db1", vp), ("dg2", vp), ("db2", vp), ("dw", vp), ("dcbias", vp), ("dcscale", vp),
        ("parts", vp),
    ]


class PackDesc(C.Structure):
    _fields_ = [("src", vp), ("dst", vp), ("dstT", vp), ("rows", i32), ("cols", i32), ("ldT", i32), ("pad", i32)]


_SIGS = {
    "qavit_version": (i32, []),
    "qavit_last_error": (C.c_char_p, []),
    "qavit_gemm_nt": (i32, [C.POINTER(GemmArgs), vp]),
    "qavit_gemm_nt_a2_supported": (i32, [i32, i32, i32, i32, i32]),
    "qavit_gemm_nt_lnbwd_supported": (i32, [i32, i32, i32, i32, i32]),
    "qavit_gemm_nt_lnbwd_parts": (i32, [i32, i32, i32]),
    "qavit_gemm_nt_grouped": (i32, [C.POINTER(GemmArgs), i32, vp]),
    "qavit_gemm_tn": (i32, [C.POINTER(GemmTnArgs), vp]),
    "qavit_gemm_tn_grouped": (i32, [C.POINTER(GemmTnArgs), i32, vp]),
    "qavit_gemm_tn_grouped_ws": (i32, [C.POINTER(GemmTnArgs), i32, vp, C.c_size_t, vp]),
    "qavit_gemm_tn_ws_bytes": (C.c_size_t, []),
    "qavit_layernorm_fwd": (i32, [i32, vp, vp, vp, vp, f32, i32, i32, vp, vp, vp, i32, i32, vp]),
    "qavit_row_stats": (i32, [i32, vp, f32, i32, i32, vp, vp, vp]),
    "qavit_row_stats_multi": (i32, [i32, i32, vp, f32, i32, i32, vp, vp, vp]),
    "qavit_layernorm_bwd_multi": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "qavit_layernorm_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, i32, vp, i32, vp, vp, vp]),
    "qavit_layernorm_bwd_sum": (i32, [i32, i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "qavit_layernorm_bwd_lin": (i32, [i32, vp, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp, vp]),
    "qavit_layernorm_bwd_lin_supported": (i32, [i32, i32, i32]),
    "qavit_layernorm_bwd_parts": (i32, [i32, i32]),
    "qavit_branch_bwd_parts": (i32, [i32, i32]),
    "qavit_cga_supported": (i32, [i32, i32, i32, i32, i32]),
    "qavit_cga_fwd": (i32, [vp, vp]),
    "qavit_cga_bwd_parts": (i32, [i32, i32]),
    "qavit_compress_fuse_supported": (i32, [i32, i32, i32, i32]),
    "qavit_compress_fuse_fwd": (i32, [vp, vp]),
    "qavit_compress_fuse_bwd_parts": (i32, [i32]),
    "qavit_compress_fuse_bwd": (i32, [vp, vp]),
    "qavit_cga_bwd": (i32, [vp, vp]),
    "qavit_mlp2_supported": (i32, [i32, i32]),
    "qavit_mlp2_fwd": (i32, [vp, vp]),
    "qavit_mlp2_bwd": (i32, [vp, vp]),
    "qavit_ccf_bwd_parts": (i32, [i32]),
    "qavit_branch_bwd": (i32, [vp, vp]),
    "qavit_ln_param_reduce": (i32, [vp, i32, vp]),
    "qavit_attn_fwd": (i32, [C.POINTER(AttnArgs), vp]),
    "qavit_attn_bwd": (i32, [C.POINTER(AttnArgs), vp]),
    "qavit_attn_ws_floats": (i64, [C.POINTER(AttnArgs)]),
    "qavit_nan_guard": (i32, [i32, vp, i64, vp, vp]),
    "qavit_branch_supported": (i32, [i32, i32, i32, i32, i32, i32, i32, i32]),
    "qavit_branch_fwd": (i32, [C.POINTER(BranchArgs), vp]),
    "qavit_tl_supported": (i32, [i32, i32, i32, i32]),
    "qavit_tl_fwd": (i32, [C.POINTER(TlArgs), vp]),
    "qavit_tl_bwd_parts": (i32, [i32, i32, i32]),
    "qavit_tl_bwd": (i32, [C.POINTER(TlBwdArgs), vp]),
    "qavit_tokmix_fwd": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "qavit_tokmix_bwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "qavit_upmix_fwd": (i32, [i32, vp, vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "qavit_upmix_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "qavit_upmix_bwd_p": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp]),
    "qavit_upmix_bwd_parts": (i32, [i32, i32, i32, i32, i32]),
    "qavit_mix3_ln_supported": (i32, [i32, i32]),
    "qavit_mix3_ln_bwd_parts": (i32, [i32, i32]),
    "qavit_gate_mix3_ln_fwd": (i32, [i32, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, vp]),
    "qavit_gate_mix3_ln_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "qavit_mix3_ln_fwd": (i32, [i32, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, vp]),
    "qavit_mix3_ln_bwd": (i32, [i32, vp, vp, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp, vp]),
    "qavit_upmix_bwd_sa_supported": (i32, [i32, i32, i32, i32]),
    "qavit_upmix_fwd_sa_supported": (i32, [i32, i32, i32, i32]),
    "qavit_upmix_fwd_sa": (i32, [i32, vp, vp, vp, f32, i32, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp, i32, i32, i32, i32, vp]),
    "qavit_upmix_bwd_sa": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, f32, i32, vp, vp]),
    "qavit_gather_pool_fwd": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "qavit_gather_pool_bwd": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "qavit_gather_pool_bwd_ld": (i32, [i32, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "qavit_ccf_mid_fwd": (i32, [C.POINTER(CcfArgs), vp]),
    "qavit_ccf_mid_bwd": (i32, [C.POINTER(CcfArgs), vp]),
    "qavit_dwconv_fwd": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "qavit_dwconv_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
    "qavit_dwconv_fwd_ld": (i32, [i32, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "qavit_dwconv_fwd_ld2": (i32, [i32, vp, vp, vp, vp, i32, vp, i32, i32, i32, i32, i32, i32, vp]),
    "qavit_dwconv_bwd_ld": (i32, [i32, vp, i32, vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "qavit_im2col": (i32, [i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "qavit_im2col_ld": (i32, [i32, vp, i32, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    "qavit_col2im": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp]),
    "qavit_branch_nan_fix": (i32, [i32, vp, i32, i32, C.POINTER(NanFix), vp]),
    "qavit_bank_stats_nanfix": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, C.POINTER(NanFix), vp]),
    "qavit_bank_stats": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, vp]),
    "qavit_bank_ws_floats": (i64, [i32, i32, i32, i32]),
    "qavit_bank_apply": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, f32, i32, vp, i32, vp, vp, vp]),
    "qavit_patchify": (i32, [i32, vp, vp, i32, i32, i32, i32, i32, vp]),
    "qavit_token_mean_fwd": (i32, [i32, vp, vp, i32, i32, i32, vp]),
    "qavit_token_mean_bwd": (i32, [i32, vp, vp, i32, i32, i32, vp]),
    "qavit_hybrid_fuse_fwd": (i32, [i32, vp, vp, vp, i32, i32, i32, vp]),
    "qavit_hybrid_fuse_bwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp, vp]),
    "qavit_sum_k": (i32, [i32, vp, i32, vp, i64, vp]),
    "qavit_rand_perm": (i32, [vp, i32, vp, i32, vp]),
    "qavit_mix_apply": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp]),
    "qavit_gate_mix_fwd": (i32, [i32, vp, vp, vp, vp, i64, vp]),
    "qavit_gate_mix_bwd": (i32, [i32, vp, vp, vp, vp, vp, i64, vp]),
    "qavit_mix2_fwd": (i32, [i32, vp, vp, vp, vp, i64, vp]),
    "qavit_mix2_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, i64, vp, vp, vp]),
    "qavit_mix3_fwd": (i32, [i32, vp, vp, vp, vp, vp, i64, f32, i32, vp, vp]),
    "qavit_mix3_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, f32, i32, vp, vp, vp, vp]),
    "qavit_scale_add_fwd": (i32, [i32, vp, vp, vp, vp, i32, i32, f32, i32, i32, vp, vp]),
    "qavit_scale_add_bwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, f32, i32, i32, vp, vp, vp, vp]),
    "qavit_chan_scale_add_fwd": (i32, [i32, vp, vp, vp, vp, i32, i32, f32, i32, i32, vp, vp]),
    "qavit_chan_scale_add_bwd": (i32, [i32, vp, vp, vp, vp, vp, i32, i32, f32, i32, i32, vp, vp]),
    "qavit_bn_fwd": (i32, [i32, vp, vp, i32, i32, vp, vp, vp, vp, f32, f32, i32, vp, vp, vp, i32, i32, i64, vp]),
    "qavit_bn_bwd": (i32, [i32, vp, vp, i32, i32, vp, vp, vp, vp, i32, i32, vp, vp, vp, vp, i32, i64, vp, vp]),
    "qavit_spatial_ln_fwd": (i32, [i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, f32, vp]),
    "qavit_spatial_ln_bwd": (i32, [i32, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp]),
    "qavit_dropout": (i32, [i32, vp, vp, i64, f32, i32, vp, vp]),
    "qavit_pack_weights": (i32, [i32, vp, i32, i32, vp]),
    "qavit_rng_advance": (i32, [vp, vp]),
    "qavit_stamp": (i32, [vp, vp]),
    "qavit_adamw": (i32, [vp, vp, vp, vp, vp, i64, vp, f32, f32, f32, f32, vp, vp, f32, vp]),
    "qavit_l2norm": (i32, [vp, i64, vp, vp, vp]),
    "qavit_ce_label_smooth": (i32, [i32, vp, vp, vp, vp, f32, i32, i32, vp, vp, vp, vp]),
    "qavit_local_clip": (i32, [vp, vp, i32, f32, vp, vp]),
    "qavit_copy2": (i32, [vp, vp, vp, vp, i64, vp]),
}

# every symbol include/qavit.h declares (checked by tests/test_abi.py against the header text)
EXPORTS = tuple(_SIGS)

_lib = None


def load():
    """dlopen the library (once).  Raises RuntimeError with build instructions if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: the HIP extension is required (no CPU fallback). "
            "Build it with `python -c 'import __graft_entry__ as g; g.build()'` or `python qa-vit_amd/build.py`.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().qavit_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")
