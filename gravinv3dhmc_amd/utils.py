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


_FMT_BUF = [None, 0]


def _format_rows(a, emit):
    import ctypes as C
    import numpy as np
    from . import _lib
    lib = _lib.load()
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)          # savetxt writes a 1-D array one value per line
    a = np.ascontiguousarray(a)
    cap = 24 * a.shape[1] + 64
    if _FMT_BUF[1] < cap:             # one scratch buffer, reused (zero-filling 12 MB per row costs as
        _FMT_BUF[0] = C.create_string_buffer(cap)     # much as formatting it)
        _FMT_BUF[1] = cap
    buf = _FMT_BUF[0]
    for row in a:
        n = lib.gh_format_row_fixed8(_lib.ptr(row), row.shape[0], buf, _FMT_BUF[1])
        if n < 0:                      # huge magnitudes: let numpy do it
            import io
            s = io.BytesIO()
            np.savetxt(s, row.reshape(1, -1), fmt='%.8f', delimiter=' ')
            emit(s.getvalue())
        else:
            emit(memoryview(buf)[:n])


def write_rows_fixed8(f, a):
    """What np.savetxt(f, a, fmt='%.8f', delimiter=' ') writes for a 1-D or 2-D array, to the
    binary file object f, formatted by the library (gh_format_row_fixed8: ~11 ns per value
    against ~280 for savetxt; the reference's model.dat / misfit.dat rows, hmc.py:241-249)."""
    _format_rows(a, f.write)


def format_rows_fixed8(a):
    """The same as bytes."""
    out = []
    _format_rows(a, lambda b: out.append(bytes(b)))
    return b"".join(out)
