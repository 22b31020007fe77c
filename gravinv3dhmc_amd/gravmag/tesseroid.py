"""gz of tesseroids (spherical prisms) on the GPU.

Mirror of the reference's `gravmag.tesseroid.gz` (gravmag/tesseroid.py:421-431 -> _dispatcher
:156-186 -> _forward_model :189-232 -> _tesseroid_numba.gz, _tesseroid_numba.py:32-71): same
arguments and return `(result, kernel2d)`; adaptive 2x2x2 Gauss-Legendre quadrature with the
distance-size ratio `ratio` (1.6 for gz) and a 100-entry subdivision stack.
"""
import warnings

import numpy as np

from .. import _lib
from ..engine import Engine
from ._common import active_cells

RATIO_G = 1.6      # tesseroid.py:77
STACK_SIZE = 100   # tesseroid.py:79

_WARN_DIVIDE = ("Stopped dividing a tesseroid because it's dimensions would be below the minimum "
                "numerical threshold (1e-6 degrees or 1e-3 m). Will compute without division. "
                "Cannot guarantee the accuracy of the solution.")
_WARN_SMALL = ("Encountered tesseroid with dimensions smaller than the numerical threshold "
               "(1e-6 degrees or 1e-3 m). Ignoring this tesseroid.")


def _valid_cells(bounds, rho):
    """tesseroid.py:126-153: assert w<=e, s<=n, top>=bottom; drop degenerate cells with a warning."""
    w, e, s, n, top, bottom = bounds.T
    bad = ~((w <= e) & (s <= n) & (top >= bottom))
    if bad.any():
        raise AssertionError("Invalid tesseroid dimensions {}".format(list(bounds[np.flatnonzero(bad)[0]])))
    tiny = ((e - w) <= 1e-6) | ((n - s) <= 1e-6) | ((top - bottom) <= 1e-3)
    if tiny.any():
        warnings.warn(_WARN_SMALL, RuntimeWarning)
    return tiny


def build_engine(lon, lat, height, model, dens=None, ratio=RATIO_G, device=0):
    """Assemble the kernel of the non-degenerate cells; returns (engine, densities, n_dropped)."""
    lon, lat, height = (np.asarray(a, dtype=np.float64) for a in (lon, lat, height))
    assert lon.shape == lat.shape == height.shape, "Input coordinate arrays must have same shape"
    assert ratio > 0, "Invalid ratio {}. Must be > 0.".format(ratio)
    bounds, rho, _ = active_cells(model, dens)
    if bounds.shape[0] == 0:
        raise ValueError("mesh has no cell with a 'density' property (and no dens given)")
    tiny = _valid_cells(bounds, rho)
    if tiny.all():
        raise ValueError("every tesseroid is below the numerical size threshold")
    keep = ~tiny
    eng = Engine(lon.size, int(keep.sum()), device=device)
    eng.set_obs(lon, lat, height)
    eng.set_cells(bounds[keep], _lib.CELL_TESSEROID, ratio)
    eng.build_G()
    if eng.kernel_stats()["warn_cells"] > 0:
        warnings.warn(_WARN_DIVIDE, RuntimeWarning)
    return eng, rho[keep], int(tiny.sum())


def gz(lon, lat, height, model, dens=None, ratio=RATIO_G, njobs=1, pool=None,
       return_kernel=True, device=0):
    """Radial (z down) gravity of the tesseroid model and its sensitivity matrix, mGal.

    Degenerate cells (tesseroid.py:139-147) are skipped by the reference's loop WITHOUT
    advancing its column counter, so the kept cells fill the leading columns of kernel2d and
    one all-zero column per skipped cell trails; the same shape is returned here."""
    assert njobs > 0, "Invalid number of jobs {}. Must be > 0.".format(njobs)
    eng, rho, ndrop = build_engine(lon, lat, height, model, dens, ratio, device)
    try:
        result = eng.forward(rho)
        kernel2d = None
        if return_kernel:
            kernel2d = eng.download_G()
            if ndrop:
                kernel2d = np.asfortranarray(
                    np.hstack([kernel2d, np.zeros((kernel2d.shape[0], ndrop))]))
    finally:
        eng.close()
    return result, kernel2d
