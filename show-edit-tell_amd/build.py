"""Build csrc/*.hip into csrc/libset_hip.so for gfx950 with hipcc (in-tree; the .so travels to
the GPU box with the gpurun snapshot).  Cross-compiles without a GPU.

    python -m show_edit_tell_amd.build [--force]
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libset_hip.so")
# the EXPERIMENTAL variant (never the product path, never bench.py's `value`): the same sources with
# -DSET_EXPERIMENTAL_GEMMS, i.e. plus csrc/experimental/*.inc (the bf16-split emulated-fp32 GEMM and the GEMM variants that lost
# their A/B).  A process loads it instead of the shipped library only when SET_LIB_VARIANT=exp is set (_lib.py).
LIB_EXP = os.path.join(CSRC, "libset_hip_exp.so")
ARCH = "gfx950"
# per-file flags.  gemm_f32.hip: the eight leading scalar kernel arguments (task count + first-workgroup table + the two row-gate pointers) are preloaded
# into SGPRs by the command processor instead of being fetched from the kernarg segment by every workgroup
FILE_FLAGS = {"gemm_f32.hip": ["-mllvm", "-amdgpu-kernarg-preload-count=8"]}


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(lib=LIB):
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(CSRC, "*.inc")) + glob.glob(os.path.join(CSRC, "experimental", "*.inc")) + [os.path.join(HERE, "..", "include", "set_hip.h")]
    return any(os.path.getmtime(p) > t for p in deps)


def build(force: bool = False, verbose: bool = False, variant: str = "") -> str:
    LIB = LIB_EXP if variant == "exp" else globals()["LIB"]
    if not force and not _stale(LIB):
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    bdir = "build_exp" if variant == "exp" else "build"
    os.makedirs(os.path.join(CSRC, bdir), exist_ok=True)
    for src in sources():
        obj = os.path.join(CSRC, bdir, os.path.basename(src) + ".o")
        objs.append(obj)
        # SET_HIPCC_FLAGS: extra compile flags (A/B experiments with -D switches on the GPU box)
        extra = os.environ.get("SET_HIPCC_FLAGS", "").split()
        if variant == "exp":
            extra = extra + ["-DSET_EXPERIMENTAL_GEMMS"]
        cmd = ([hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-comment"] + extra +
               FILE_FLAGS.get(os.path.basename(src), []) + ["-c", src, "-o", obj])
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout.decode())
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, variant="exp" if "--exp" in sys.argv else ""))
