"""Compile libgravhmc.so (hand-written HIP for gfx950) in-tree with hipcc."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "gravhmc.hip")
_CSRC = os.path.join(_HERE, "csrc")
DEPS = sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".h"))) + \
       [os.path.join(os.path.dirname(_HERE), "include", "gravhmc.h")]
LIB = os.path.join(_HERE, "libgravhmc.so")


def hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libgravhmc.so)")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS if os.path.exists(d))


def build(force=False, verbose=False):
    """Build the shared library if it is missing or older than its sources."""
    from . import isa_check
    if not force and not stale():
        if not os.path.exists(isa_check.STAMP):
            isa_check.stamp(hipcc())
        return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-Wno-unused-value", SRC, "-o", LIB + ".tmp", "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    os.replace(LIB + ".tmp", LIB)
    # the generated code of the kernel with hand-counted waits, checked with the compiler that built it
    bad = isa_check.stamp(hipcc())
    if bad:
        print("gravinv3dhmc_amd.build: batch_team_kernel's generated code has findings (the team form of the "
              "stored-kernel batch stays off):\n  " + "\n  ".join(bad[:5]))
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
