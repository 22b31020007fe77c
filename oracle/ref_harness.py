"""Import the reference implementation from /root/reference (build container ONLY).

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py to pin the oracle and to generate
the vectors under tests/golden/.  Nothing on the GPU box may import this module: the
reference does not travel.  No reference source is copied; the modules are imported from
where they lie, with the shims SURVEY.md 8c lists:

  * numpy.float = float         (removed alias used by _prism.pyx / prism.py / utils.py)
  * sys.modules["numba"]        identity `jit` (numba is not installed; the very same
                                source then runs as plain Python)
  * sys.modules["pywt"]         empty stub (PyWavelets is not installed; only needed so that
                                gravmag/__init__.py imports -- the wavelet path is NOT
                                runnable through the reference here)
  * sys.modules["mpi4py"]       not needed (drivers are not imported)
  * gravmag.__path__ += oracle/_ref   so `from . import _prism` finds the extension built
                                by oracle/build_ref.py from the reference's own .pyx
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get("GRAVHMC_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def available():
    return os.path.isdir(os.path.join(REF, "gravmag"))


def load():
    """Returns a namespace with the reference modules (prism, tesseroid, mesher, potential, hmc)."""
    if not available():
        raise RuntimeError("reference not present at %s" % REF)
    if not hasattr(np, "float"):
        np.float = float
    if "numba" not in sys.modules:
        nb = types.ModuleType("numba")

        def jit(*a, **k):
            if len(a) == 1 and callable(a[0]) and not k:
                return a[0]
            return lambda f: f

        nb.jit = jit
        nb.njit = jit
        sys.modules["numba"] = nb
    if "pywt" not in sys.modules:
        sys.modules["pywt"] = types.ModuleType("pywt")
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import matplotlib

    matplotlib.use("agg")
    # make `gravmag._prism` resolvable without touching the read-only tree
    from oracle import build_ref

    build_ref.main()
    import importlib.machinery
    import importlib.util

    pkg = types.ModuleType("gravmag")
    pkg.__path__ = [os.path.join(REF, "gravmag"), os.path.join(_HERE, "_ref")]
    pkg.__package__ = "gravmag"
    sys.modules.setdefault("gravmag", pkg)
    import gravmag.prism as prism
    import gravmag.tesseroid as tesseroid

    assert prism._prism is not None, "reference Cython extension did not import"
    import mesher
    import constants
    # inversion/potential.py imports vis (matplotlib only; mayavi is lazy) and the compressors
    import gravmag.compressor1D  # noqa: F401  (uses the pywt stub; import only)
    import gravmag.compressor3D  # noqa: F401
    sys.modules["gravmag"].prism = prism
    sys.modules["gravmag"].tesseroid = tesseroid
    sys.modules["gravmag"].compressor1D = sys.modules["gravmag.compressor1D"]
    sys.modules["gravmag"].compressor3D = sys.modules["gravmag.compressor3D"]
    from inversion import potential, hmc

    return types.SimpleNamespace(prism=prism, tesseroid=tesseroid, mesher=mesher,
                                 constants=constants, potential=potential, hmc=hmc,
                                 _prism=prism._prism)
