"""Small host helpers the reference's drivers use around the hot path.

`rho2carve` / `carve2rho` (utils.py:714-749): move between the full-mesh cell order and the
vector of active (uncarved) cells; `regular` (utils.py:114-151): observation grid with x slow,
y fast.  Vectorised restatements (the reference's `regular` no longer runs under numpy >= 2,
SURVEY section 2 #11)."""
import numpy as np


def rho2carve(rho, mask):
    """Drop the carved cells (flat indices in `mask`) from a full-mesh vector."""
    rho = np.asarray(rho)
    keep = np.ones(rho.shape[0], dtype=bool)
    if len(mask):
        keep[np.asarray(mask, dtype=np.int64)] = False
    return rho[keep]


def carve2rho(rho_carved, mask, size, fill=0.0):
    """Scatter a vector of active cells back into a full-mesh vector of `size` cells; carved
    cells get `fill`."""
    out = np.full(int(size), fill, dtype=np.float64)
    keep = np.ones(int(size), dtype=bool)
    if len(mask):
        keep[np.asarray(mask, dtype=np.int64)] = False
    out[keep] = np.asarray(rho_carved, dtype=np.float64)
    return out


def regular(area, shape, z=None):
    """Regular grid over area = (x1, x2, y1, y2) with shape = (nx, ny): returns [x, y(, z)]
    raveled with x varying slowest (the order of model01_singlecube.py:93-94)."""
    x1, x2, y1, y2 = area
    nx, ny = shape
    ys, xs = np.meshgrid(np.linspace(y1, y2, ny), np.linspace(x1, x2, nx))
    out = [xs.ravel(), ys.ravel()]
    if z is not None:
        out.append(z * np.ones(nx * ny, dtype=np.float64))
    return out
