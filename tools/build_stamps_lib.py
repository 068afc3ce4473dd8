"""Diagnostic build: libqavit_stamps.so = the regular objects + ONE source recompiled with an in-kernel-stamps macro (s_memtime at phase
boundaries, read back by tools/branch_stamps.py / tools/token_stamps.py through QAVIT_LIB=qa-vit_amd/libqavit_stamps.so).  Never the product
library.   usage: build_stamps_lib.py [branch_fwd QAVIT_BRANCH_STAMPS]   |   build_stamps_lib.py tokens_bf16 QAVIT_TOKEN_STAMPS"""
import glob, os, subprocess, sys
src = sys.argv[1] if len(sys.argv) > 1 else "branch_fwd"
macro = sys.argv[2] if len(sys.argv) > 2 else "QAVIT_BRANCH_STAMPS"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qa-vit_amd")
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result"] + ["-D" + m for m in macro.split(",")]
os.makedirs("/tmp/stampbuild", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", f"{root}/csrc/{src}.hip", "-o", f"/tmp/stampbuild/{src}.o"])
objs = [o for o in glob.glob(root + "/build/*.o") if not o.endswith(f"/{src}.o")] + [f"/tmp/stampbuild/{src}.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", root + "/libqavit_stamps.so"])
print(root + "/libqavit_stamps.so")
