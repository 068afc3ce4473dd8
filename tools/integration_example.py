"""The reference-side binding INTEGRATION.md shows, as a runnable script: the reference's own `efficient_attention`
(HQAViT_CIFAR100.py:355-397) for the cross-attention branch (:613-626) on the HIP kernel, bound with nothing but ctypes
and checked against F.scaled_dot_product_attention.  Run on a GPU box:  python tools/integration_example.py"""
import ctypes
import os

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "qa-vit_amd", "libqavit_hip.so"))
lib.qavit_last_error.restype = ctypes.c_char_p

i32, i64, f32, vp = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p


class AttnArgs(ctypes.Structure):                  # field order = struct qavit_attn_args in include/qavit.h
    _fields_ = [("dtype", i32), ("mode", i32),
                ("G", i32), ("Nq", i32), ("L", i32), ("H", i32), ("D", i32), ("KC", i32), ("S", i32),
                ("groups_per_b", i32), ("q_rows_per_b", i32), ("k_rows_per_b", i32), ("q_tbl", vp), ("k_tbl", vp),
                ("q", vp), ("ldq", i64), ("k_tok", vp), ("ldk", i64), ("v_tok", vp), ("ldv", i64),
                ("E_k", vp), ("E_v", vp), ("sh_k", vp), ("sh_v", vp), ("o", vp), ("ldo", i64), ("nan_flag", vp),
                # backward operands (unused by qavit_attn_fwd)
                ("d_o", vp), ("lddo", i64), ("dq", vp), ("lddq", i64), ("dk_tok", vp), ("lddk", i64), ("dv_tok", vp), ("lddv", i64),
                ("ws", vp), ("ws_floats", i64), ("dE_k", vp), ("dE_v", vp), ("dsh_k", vp), ("dsh_v", vp),
                # attention-probability dropout (the dropout_p handed to SDPA): off here
                ("drop_p", f32), ("drop_site", i32), ("rng", vp)]


def cross_attention_core(q_rows: torch.Tensor, k_bank_proj: torch.Tensor, v_bank_proj: torch.Tensor, batch: int, heads: int) -> torch.Tensor:
    """softmax(q K^T / sqrt(D)) V per (image, head) with K, V = the bank's projections, shared by every image.
    q_rows [B*N, C] bf16 (row-major token matrix), k/v [S, C] fp32 -> [B*N, C] bf16."""
    rows, C = q_rows.shape
    N = rows // batch
    a = AttnArgs(dtype=1, mode=1, G=batch, Nq=N, L=0, H=heads, D=C // heads, S=k_bank_proj.shape[0])
    out = torch.empty_like(q_rows)
    a.q, a.ldq = q_rows.data_ptr(), C
    a.sh_k, a.sh_v = k_bank_proj.data_ptr(), v_bank_proj.data_ptr()
    a.o, a.ldo = out.data_ptr(), C
    rc = lib.qavit_attn_fwd(ctypes.byref(a), vp(torch.cuda.current_stream().cuda_stream))
    if rc:
        raise RuntimeError(lib.qavit_last_error().decode())
    return out


if __name__ == "__main__":
    B, N, C, H, S = 8, 16, 192, 4, 16
    g = torch.Generator().manual_seed(0)
    q = torch.randn(B * N, C, generator=g).cuda().to(torch.bfloat16)
    k, v = torch.randn(S, C, generator=g).cuda(), torch.randn(S, C, generator=g).cuda()
    got = cross_attention_core(q, k, v, B, H).float()
    qh = q.float().view(B, N, H, C // H).transpose(1, 2)
    kh = k.view(1, S, H, C // H).transpose(1, 2).expand(B, -1, -1, -1)
    vh = v.view(1, S, H, C // H).transpose(1, 2).expand(B, -1, -1, -1)
    ref = torch.nn.functional.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B * N, C)
    err = float((got - ref).abs().max() / ref.abs().max())
    print(f"max-rel error vs F.scaled_dot_product_attention: {err:.2e}")
    assert err < 3e-2
