"""Build libpswin_hip.so (the C-ABI library of include/pswin.h) in-tree with hipcc for gfx950.

    python -m panoswintransformerobjectdetection_amd.build [--force]

hipcc cross-compiles without a GPU.  The .so is git-ignored but travels with the gpurun snapshot.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB = os.path.join(PKG, "libpswin_hip.so")
# A/B builds (tools/ab_*.sh): PSWIN_BUILD_FLAGS="-DX=1 ..." adds compiler flags, PSWIN_BUILD_OUT=<path> names the library (objects go
# next to it); the default build ignores both
EXTRA_FLAGS = os.environ.get("PSWIN_BUILD_FLAGS", "").split()
OUT = os.environ.get("PSWIN_BUILD_OUT")
SOURCES = ["pswin_index.hip", "pswin_geom.hip", "pswin_move.hip", "pswin_attn.hip", "pswin_norm.hip", "pswin_bn.hip", "pswin_stem.hip", "pswin_mlp.hip", "pswin_gemm.hip", "pswin_gemm_nt.hip", "pswin_gemm_tn.hip", "pswin_fused.hip", "pswin_qkvattn.hip", "pswin_optim.hip", "pswin_roi.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-function",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "pswin.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    lib = OUT or LIB
    if not OUT and not force and not _stale():
        return LIB
    objs = []
    procs = []
    objdir = CSRC
    if OUT:
        objdir = OUT + ".objs"
        os.makedirs(objdir, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        cmd = [HIPCC] + FLAGS + (EXTRA_FLAGS if OUT else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if out.strip() and verbose:
            print(out)
        if p.returncode:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print("built", LIB)
