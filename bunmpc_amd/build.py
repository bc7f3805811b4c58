"""Builds libbunmpc_hip.so (gfx950 kernels + C-ABI) in-tree with hipcc.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with
the working tree to the GPU box."""
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["biconvex_admm.hip", "bunmpc_capi.hip", "ik_ddp.hip", "bunmpc_ik_capi.hip"]
HEADERS = [os.path.join(CSRC, "biconvex_kernels.h"), os.path.join(CSRC, "ik_types.h"), os.path.join(CSRC, "rbd_device.h"),
           os.path.join(os.path.dirname(_HERE), "include", "bunmpc.h")]
LIB = os.path.join(_HERE, "libbunmpc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared"]


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=()):
    if not force and not is_stale():
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s: cannot build %s" % (HIPCC, LIB))
    cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else [])
    print(LIB)
