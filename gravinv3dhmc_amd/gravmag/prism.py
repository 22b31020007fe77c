"""gz of right rectangular prisms on the GPU.

Mirror of the reference's `gravmag.prism.gz` (gravmag/prism.py:911-918 -> _dispatcher_gravity
:998-1038 -> _gz :291-316 -> _prism.gz, _prism.pyx:265-290): same arguments, same return
`(result, kernel2d)` in mGal for densities in g/cm^3.  `njobs`/`pool` are accepted and
ignored (the reference uses them for host multiprocessing; the assembly here is one HIP
launch over all (observation, cell) pairs).
"""
import numpy as np

from .. import _lib
from ..engine import Engine
from ._common import active_cells


def build_engine(xp, yp, zp, prisms, dens=None, device=0):
    """Assemble the dense kernel for `prisms` on the device; returns (engine, densities)."""
    xp, yp, zp = (np.asarray(a, dtype=np.float64) for a in (xp, yp, zp))
    if xp.shape != yp.shape or xp.shape != zp.shape:
        raise ValueError("Input arrays xp, yp, and zp must have same length!")
    bounds, rho, _ = active_cells(prisms, dens)
    if bounds.shape[0] == 0:
        raise ValueError("mesh has no cell with a 'density' property (and no dens given)")
    eng = Engine(xp.size, bounds.shape[0], device=device)
    eng.set_obs(xp, yp, zp)
    eng.set_cells(bounds, _lib.CELL_PRISM)
    eng.build_G()
    return eng, rho


def gz(xp, yp, zp, prisms, dens=None, njobs=1, pool=None, return_kernel=True, device=0):
    """Vertical gravity of the prism model and its sensitivity matrix.

    Returns (result[N], kernel2d[N, M_active]); kernel2d is Fortran-ordered.  Pass
    return_kernel=False to skip the device->host copy of the matrix (returns None)."""
    eng, rho = build_engine(xp, yp, zp, prisms, dens, device)
    try:
        result = eng.forward(rho)
        kernel2d = eng.download_G() if return_kernel else None
    finally:
        eng.close()
    return result, kernel2d
