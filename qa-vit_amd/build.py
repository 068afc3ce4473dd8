"""Build libqavit_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Each .hip file is compiled to an object (in parallel, cached on source mtime) and linked into
``qa-vit_amd/libqavit_hip.so`` -- a plain C-ABI shared library (include/qavit.h), no torch headers.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libqavit_hip.so")
OBJ = os.path.join(HERE, "build")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wno-unused-result", "-Rpass-analysis=kernel-resource-usage"]
FLAGS += os.environ.get("QAVIT_EXTRA_HIPCC_FLAGS", "").split()       # diagnostic builds, e.g. -DQAVIT_BRANCH_STAMPS (tools/branch_stamps.py)

# A spill is HBM traffic and a dependent round trip per access: no kernel of a training / inference step may use scratch memory.  The
# build reads hipcc's per-kernel resource remarks and FAILS on a non-zero ScratchSize unless the (mangled) kernel name contains one of
# these strings -- instantiations no shipped configuration launches:
ALLOW_SCRATCH = (
    "dwconv_bwd8_kernel",                 # the one-lane-per-channel 8x8 backward: superseded by dwconv_bwdt (kept for tools/bench_dw.py)
    "layernorm_bwd_v4_kernelIfLi2", "layernorm_bwd_v4_kernelIfLi4", "layernorm_bwd_v4_kernelIDF16bLi2", "layernorm_bwd_v4_kernelIDF16bLi4",
    "layernorm_bwd_v4_multi_kernelIfLi2",  # LayerNorm backward over 257..1024 channels: every LayerNorm of the models has C <= 256
    "bank_stats_kernel",                  # fp32 parity path of the chunked (224 px) bank statistics
    "sln_bwd_kernel",                     # spatial LayerNorm backward of the v2 stem at 64 rows per thread (fp32 / bf16): v2-stem variant only
)
# Step kernels allowed a BOUNDED amount of scratch, with the reason.  (name fragment, max bytes per lane)
ALLOW_SCRATCH_UP_TO = (
    # the one-launch weight-gradient kernel holds twenty tile-class bodies in one problem loop; twelve lane-invariant registers (thread
    # geometry that lives across the loop) are stored ONCE when a workgroup enters the problem loop (ten scratch stores, loop depth 1) and
    # a class body reloads the one to four it needs in its prologue (loop depth 2 = once per (problem, tile) segment, ~2 segments per
    # workgroup and launch); none inside a chunk loop (depth 3).  Read off the ISA: `hipcc -S --cuda-device-only`, every scratch_
    # instruction of the kernel sits in a block the compiler annotates Depth=1 or Depth=2.
    ("gemm_tn_uni_kernel", 48),
)


def _scratch_report(stderr: str):
    """-> [(kernel, bytes_per_lane, vgprs)] from -Rpass-analysis=kernel-resource-usage remarks."""
    out, name, vg = [], None, 0
    for line in stderr.splitlines():
        if "Function Name:" in line:
            name = line.split("Function Name:")[1].split("[-R")[0].strip()
        elif "    VGPRs:" in line and "remark" in line:
            vg = int(line.split("VGPRs:")[1].split("[")[0])
        elif "ScratchSize [bytes/lane]:" in line and name:
            out.append((name, int(line.split("ScratchSize [bytes/lane]:")[1].split("[")[0]), vg))
    return out


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.exists(c) or c == "hipcc":
            return c


def _deps_mtime():
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".cuh", ".h"))]
    hdrs.append(os.path.join(HERE, "..", "include", "qavit.h"))
    return max(os.path.getmtime(h) for h in hdrs)


def build(verbose: bool = False, force: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_m):
            jobs.append([_hipcc(), *FLAGS, "-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        rep = _scratch_report(r.stderr)
        bad = [(k, b) for k, b, _ in rep if b > 0 and not any(a in k for a in ALLOW_SCRATCH)
               and not any(a in k and b <= cap for a, cap in ALLOW_SCRATCH_UP_TO)]
        if bad:
            if "-o" in cmd and os.path.exists(cmd[cmd.index("-o") + 1]):
                os.remove(cmd[cmd.index("-o") + 1])            # do not let a later build link the spilling object
            raise RuntimeError("kernels with scratch memory (spills) in " + cmd[-3] + ": " + ", ".join(f"{k}: {b} B/lane" for k, b in bad))
        if "-c" in cmd:                                        # per-kernel registers / scratch beside the object (tools read it)
            with open(cmd[cmd.index("-o") + 1][:-2] + ".res", "w") as f:
                f.writelines(f"{k} vgprs={v} scratch={b}\n" for k, b, v in rep)
        if verbose:
            other = "\n".join(l for l in r.stderr.splitlines() if "warning:" in l or "error:" in l)
            if other.strip():
                print(other, file=sys.stderr)

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            list(ex.map(run, jobs))
    if jobs or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(verbose=True, force="--force" in sys.argv))
