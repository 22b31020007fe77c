"""Compile the reference's own Cython prism kernel where it lies (build container only).

TEST INFRASTRUCTURE ONLY.  Reads /root/reference/gravmag/_prism.pyx (never copied into the
repository), runs Cython + gcc on it and leaves exactly one artefact:
    oracle/_ref/_prism.<abi>.so
(`oracle/_ref/` is git-ignored; the intermediate C file is deleted).  The reference's own
setup.py (gravmag/setup.py) is not used: it is a Windows /openmp recipe.
"""
import os
import subprocess
import sys
import sysconfig

import numpy

REF = os.environ.get("GRAVHMC_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_ref")


def main():
    pyx = os.path.join(REF, "gravmag", "_prism.pyx")
    if not os.path.exists(pyx):
        print("reference not present at", REF, "- skipping oracle/_ref build")
        return 0
    os.makedirs(OUT, exist_ok=True)
    suffix = sysconfig.get_config_var("EXT_SUFFIX")
    so = os.path.join(OUT, "_prism" + suffix)
    if os.path.exists(so) and os.path.getmtime(so) >= os.path.getmtime(pyx):
        return 0
    cfile = os.path.join(OUT, "_prism_generated.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "-o", cfile, pyx])
    inc = sysconfig.get_paths()["include"]
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w", "-ffp-contract=off",
                           "-DNPY_NO_DEPRECATED_API=NPY_1_7_API_VERSION",
                           "-I", inc, "-I", numpy.get_include(), cfile, "-o", so, "-lm"])
    os.remove(cfile)
    print("built", so)
    return 0


if __name__ == "__main__":
    sys.exit(main())
