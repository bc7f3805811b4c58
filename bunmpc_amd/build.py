"""Builds libbunmpc_hip.so (gfx950 kernels + C-ABI) in-tree with hipcc.

hipcc cross-compiles without a GPU; the built .so is git-ignored but travels with
the working tree to the GPU box.  Sources are compiled to objects one by one (cached under
csrc/_obj, keyed by the flags) and linked, so touching one kernel file recompiles only that file.

    python -m bunmpc_amd.build [--force] [--usage]
Environment: HIPCC, BUNMPC_EXTRA_FLAGS (extra compile flags, e.g. -DBWD_PROFILE), BUNMPC_LIB_OUT
(output path, for side-by-side experiment builds; load it with BUNMPC_LIB=<path>)."""
import contextlib
import fcntl
import hashlib
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
SOURCES = ["biconvex_admm.hip", "biconvex_admm_f32.hip", "biconvex_latency.hip", "bunmpc_capi.hip", "ik_ddp.hip", "bunmpc_ik_capi.hip", "plan_gen.hip", "id_ctrl.hip", "perturb.hip"]
HEADERS = [os.path.join(CSRC, h) for h in ("biconvex_kernels.h", "biconvex_lanes.h", "biconvex_admm_body.h", "ik_types.h", "rbd_device.h", "rbd_quad.h", "lds_batch.h", "id_types.h", "perturb_types.h")] + \
          [os.path.join(os.path.dirname(_HERE), "include", "bunmpc.h")]
LIB = os.path.join(_HERE, "libbunmpc_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
# per-file extra flags (none now: -freciprocal-math -fapprox-func on ik_ddp.hip turns its 284 IEEE fp64 divisions into v_rcp_f64 +
# Newton steps, measured gain on the MI355X: none -- the divisions sit off the chains that set the pace -- so IEEE division stays)
# ik_ddp.hip: -ffp-contract=on.  hipcc's default for device code (fast) fuses a multiply into an add across statements, in the
# back end, where the decision depends on what surrounds the expression -- and the same source, instantiated once per wave role
# and per mapping, then rounds differently here and there (seen: one problem of 4096 whose final cost differed by one ulp between
# the four-problems-per-wave line search and the role-split one).  With `on` a product is fused only with the sum of its own
# expression, decided in the front end: every instantiation of a piece of source gets the same arithmetic, so "a problem's
# result does not depend on how it was scheduled" holds by construction.
# biconvex_admm_f32.hip: no SLP vectoriser -- packed fp32 operations cost this kernel ~90 registers and its two-waves-per-SIMD
# build 40-60 values in scratch memory (the reasons and the measurement are in the file's header).
# -amdgpu-sched-strategy=max-ilp (the scheduler orders for instruction-level parallelism instead of register pressure): the
# one-problem-per-wave kernel is one long dependent chain per wave, batch-1 p50 1.49 -> 1.455 ms; the fp32 kernel 6.03 -> 5.97 ms.
# Timed and NOT taken elsewhere: the fp64 batch kernel (headline 4.095 -> 4.12 ms; its 64-lane shape 9.39 -> 9.24 ms) and ik_ddp.hip
# (derivative pass -2 %, Riccati pass and line search +1.5 %).
# -amdgpu-use-amdgpu-trackers (the scheduler follows register pressure with the target's own trackers): fp64 batch kernel, headline
# 4.095 -> 4.007 ms, its 64-lane shape 9.39 -> 8.95 ms; worse on ik_ddp.hip (Riccati pass +5 %) and on the one-problem-per-wave
# kernel (1.452 -> 1.469 ms), level on the fp32 kernel.  With it -amdgpu-disable-unclustered-high-rp-reschedule (no second scheduling
# pass against register pressure -- the kernel has 512 registers to itself): headline 4.014 -> 3.98 ms (64-lane shape 8.95 -> 9.0).
# No effect or worse on top: -amdgpu-schedule-metric-bias=0, -amdgpu-schedule-relaxed-occupancy, -amdgpu-early-ifcvt, max-ilp,
# -amdgpu-disable-clustered-low-occupancy-reschedule.
FILE_FLAGS = {"ik_ddp.hip": ["-ffp-contract=on"],
              "biconvex_admm.hip": ["-mllvm", "-amdgpu-use-amdgpu-trackers", "-mllvm", "-amdgpu-disable-unclustered-high-rp-reschedule"],
              "biconvex_admm_f32.hip": ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp"],
              "biconvex_latency.hip": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}


def _without_mllvm(cmd):
    out, skip = [], False
    for tok in cmd:
        if skip:
            skip = False
        elif tok == "-mllvm":
            skip = True
        else:
            out.append(tok)
    return out


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS + [os.path.abspath(__file__)]      # (this file: the flags live here)
    return any(os.path.getmtime(d) > t for d in deps)


@contextlib.contextmanager
def build_lock():
    """One builder at a time per tree (the ranks of a multi-GPU job start together and share csrc/_obj and the .so): the
    first to arrive builds, the others wait here and then find the library fresh."""
    os.makedirs(OBJ, exist_ok=True)
    with open(os.path.join(OBJ, ".build.lock"), "w") as f:
        fcntl.flock(f, fcntl.LOCK_EX)
        try:
            yield
        finally:
            fcntl.flock(f, fcntl.LOCK_UN)


def build(force=False, verbose=False, extra_flags=()):
    out = os.environ.get("BUNMPC_LIB_OUT", LIB)
    if not force and out == LIB and not is_stale():
        return LIB
    with build_lock():
        if not force and out == LIB and not is_stale():      # another process built it while this one waited
            return LIB
        return _build_locked(out, force, verbose, extra_flags)


def _build_locked(out, force, verbose, extra_flags):
    if not os.path.exists(HIPCC):
        raise RuntimeError("hipcc not found at %s: cannot build %s" % (HIPCC, out))
    flags = FLAGS + os.environ.get("BUNMPC_EXTRA_FLAGS", "").split() + list(extra_flags)
    key = hashlib.sha1((" ".join(flags) + repr(sorted(FILE_FLAGS.items()))).encode()).hexdigest()[:10]
    os.makedirs(OBJ, exist_ok=True)
    newest_header = max(os.path.getmtime(h) for h in HEADERS)
    objs, procs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        op = os.path.join(OBJ, "%s.%s.o" % (src, key))
        objs.append(op)
        if force or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), newest_header):
            cmd = [HIPCC] + flags + FILE_FLAGS.get(src, []) + ["-c", sp, "-o", op]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            # -mllvm options are LLVM-internal switches (scheduler tuning worth 1-3 %), not a stable interface: another hipcc may not
            # know them ("Unknown command line argument").  One retry without them; -ffp-contract=on and -fno-slp-vectorize stay --
            # they carry the bit-identity and no-scratch guarantees (tests/test_biconvex_gpu.py, tests/test_ik_gpu.py).
            plain = _without_mllvm(cmd)
            if plain == cmd:
                raise subprocess.CalledProcessError(p.returncode, cmd)
            print("bunmpc_amd.build: retrying without -mllvm options: " + " ".join(plain), file=sys.stderr)
            subprocess.check_call(plain)
    tmp = "%s.tmp.%d" % (out, os.getpid())     # linked beside, then renamed: a process loading the library never sees half of it
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    os.replace(tmp, out)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True,
                extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--usage" in sys.argv else []))
