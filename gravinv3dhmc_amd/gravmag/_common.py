"""Shared helpers of the prism / tesseroid front-ends."""
import numpy as np


def active_cells(model, dens):
    """Bounds table (M, 6) and density vector (M) of the cells a reference loop would visit.

    Mirrors the skipping rule of gravmag/prism.py:299-301 and gravmag/tesseroid.py:95-98,
    126-153: a cell is skipped when it is None (carved) or when it has no 'density' property
    and no `dens` override is given."""
    props = getattr(model, "props", None)
    if hasattr(model, "cell_bounds") and hasattr(model, "active_index"):
        if dens is None and (props is None or "density" not in props):
            return np.zeros((0, 6)), np.zeros(0), None
        bounds = model.cell_bounds(active_only=True)
        idx = model.active_index()
        if dens is not None:
            rho = np.full(len(idx), float(dens))
        else:
            rho = np.asarray(props["density"], dtype=np.float64)[idx]
        return bounds, rho, idx
    rows, rho = [], []
    for cell in model:
        if cell is None or ("density" not in cell.props and dens is None):
            continue
        rows.append(cell.get_bounds())
        rho.append(float(dens) if dens is not None else float(cell.props["density"]))
    return np.asarray(rows, dtype=np.float64).reshape(-1, 6), np.asarray(rho, dtype=np.float64), None
