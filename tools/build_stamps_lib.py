"""Diagnostic build: libqavit_stamps.so = the regular objects + branch_fwd.hip compiled with -DQAVIT_BRANCH_STAMPS (in-kernel s_memtime stamps,
read back by tools/branch_stamps.py through QAVIT_LIB=qa-vit_amd/libqavit_stamps.so).  Never the product library."""
import glob, os, subprocess
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "qa-vit_amd")
flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-Wno-unused-result", "-DQAVIT_BRANCH_STAMPS"]
os.makedirs("/tmp/stampbuild", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", *flags, "-c", root + "/csrc/branch_fwd.hip", "-o", "/tmp/stampbuild/branch_fwd.o"])
objs = [o for o in glob.glob(root + "/build/*.o") if not o.endswith("branch_fwd.o")] + ["/tmp/stampbuild/branch_fwd.o"]
subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", *objs, "-o", root + "/libqavit_stamps.so"])
print(root + "/libqavit_stamps.so")
