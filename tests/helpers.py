"""Shared problem builders for the tests (synthetic inputs of BASELINE.json's configs)."""
import numpy as np


def c1_inputs():
    """Config C1: uniformgrid singlecube, 20x30x10 prisms, 600 obs on z=0."""
    from gravinv3dhmc_amd import mesher
    mesh = mesher.PrismMesh((0, 2000, 0, 3000, 0, 1000), (100, 100, 100))
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 3000, 30), np.linspace(0, 2000, 20))]
    zp = np.zeros_like(xp)
    return mesh, xp, yp, zp


def relmax(a, b):
    a, b = np.asarray(a), np.asarray(b)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))
