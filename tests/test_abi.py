"""The C-ABI library loads on a machine without a GPU and exports every symbol include/qavit.h declares (no compute
calls here).  Also: the ctypes structures mirror the header's field order."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "qavit.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(qavit_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol(Q):
    lib = Q.lib.load()
    syms = header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/qavit.h but not exported"
    assert sorted(Q.lib.EXPORTS) == syms, set(Q.lib.EXPORTS) ^ set(syms)
    assert lib.qavit_version() >= 1
    assert isinstance(lib.qavit_last_error(), bytes)


def _struct_fields(name):
    txt = open(os.path.join(ROOT, "include", "qavit.h")).read()
    body = re.search(r"typedef struct " + name + r" \{(.*?)\} " + name + ";", txt, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        names = decl.split(",")
        first = names[0].split()
        out.append(re.sub(r"\[.*\]", "", first[-1].lstrip("*")))
        for n in names[1:]:
            out.append(re.sub(r"\[.*\]", "", n.strip().lstrip("*")))
    return out


def test_ctypes_structs_match_header(Q):
    pairs = [("qavit_gemm_args", Q.lib.GemmArgs), ("qavit_gemm_tn_args", Q.lib.GemmTnArgs), ("qavit_attn_args", Q.lib.AttnArgs),
             ("qavit_ccf_args", Q.lib.CcfArgs), ("qavit_pack_desc", Q.lib.PackDesc), ("qavit_branch_args", Q.lib.BranchArgs),
             ("qavit_branch_bwd_args", Q.lib.BranchBwdArgs), ("qavit_ln_reduce_desc", Q.lib.LnReduceDesc), ("qavit_cga_args", Q.lib.CgaArgs), ("qavit_cga_bwd_args", Q.lib.CgaBwdArgs), ("qavit_cfuse_args", Q.lib.CfuseArgs), ("qavit_cfuse_bwd_args", Q.lib.CfuseBwdArgs),
             ("qavit_mlp2_args", Q.lib.Mlp2Args), ("qavit_mlp2_bwd_args", Q.lib.Mlp2BwdArgs),
             ("qavit_tl_args", Q.lib.TlArgs), ("qavit_tl_bwd_args", Q.lib.TlBwdArgs)]
    for cname, st in pairs:
        assert [f[0] for f in st._fields_] == _struct_fields(cname), cname


def test_argument_validation_without_gpu(Q):
    """Entry points reject bad arguments before touching the device (callable on a CPU-only host)."""
    lib = Q.lib.load()
    a = Q.lib.GemmArgs()
    assert lib.qavit_gemm_nt(ctypes.byref(a), None) == -1
    assert b"null operand" in lib.qavit_last_error()
    assert lib.qavit_layernorm_fwd(0, None, None, None, None, 1e-5, 4, 8, None, None, None, 0, 0, None) == -1
    assert lib.qavit_dropout(7, None, None, 0, 0.5, 0, None, None) == -1


def test_missing_library_fails_loudly(Q, monkeypatch, tmp_path):
    monkeypatch.setattr(Q.lib, "_lib", None)
    monkeypatch.setattr(Q.lib, "LIB_PATH", str(tmp_path / "nope.so"))
    try:
        Q.lib.load()
    except RuntimeError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")
