"""Build libcavp_hip.so (gfx950) in-tree with hipcc.  `python -m cavp_amd.build [--force]`.

The library is the product's only compute path; it is compiled ahead of time (hipcc cross-compiles without a
GPU) and travels to the GPU box next to the sources, so nothing is JIT-built at run time."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcavp_hip.so")
SOURCES = ["conv_igemm.hip", "conv_igemm_big.hip", "pointwise.hip", "train_pointwise.hip", "conv_wgrad.hip", "conv_wgrad_big.hip", "contrast.hip", "pvt_ops.hip", "pvt_train.hip", "layernorm.hip", "attn_gate.hip", "attn_rank1.hip", "mel_frontend.hip", "optimizer.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-munsafe-fp-atomics", "-ffp-contract=on"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found: libcavp_hip.so cannot be built (ROCm toolchain required)")


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "cavp_hip.h"), __file__]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, profile: bool = False) -> str:
    """profile=True adds -DCAVP_PROFILE: the kernels' compile-time anatomy variants (tools/bench_conv.py --variants);
    the product library carries none of them."""
    if not force and not profile and not _stale():
        return LIB
    hipcc = _hipcc()
    flags = FLAGS + (["-DCAVP_PROFILE"] if profile else [])
    objs = []
    # the profile flavour lives beside the product: build_profile/*.o -> libcavp_hip_profile.so (tools pick it with --lib)
    objdir = os.path.join(HERE, "build_profile" if profile else "build")
    lib = os.path.join(HERE, "libcavp_hip_profile.so") if profile else LIB
    os.makedirs(objdir, exist_ok=True)
    procs = []
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "cavp_hip.h")]
    same_flavour = True
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        deps = [os.path.join(CSRC, src), __file__] + headers
        if not force and same_flavour and os.path.exists(obj) and all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps):
            continue   # incremental: this object is newer than its source and every header
        cmd = [hipcc, *flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, profile="--profile" in sys.argv))
