"""Posterior statistics of the accepted samples (what the reference's plot scripts compute).

`example/uniformgrid/plot_uniform.py:44-55,101-155`: per-cell mean and standard deviation over
the LAST `last` rows of model.dat, the forward responses of both, and the two summary numbers
RMSD = sqrt(|dobs - d(mean)|^2 / N), RMSM = sqrt(|rho_true - mean|^2 / M).  Here they come from
the ring of accepted samples the engine keeps in HBM (`Engine.posterior_window`), or from a
sample file for comparison.
"""
import numpy as np


def stats_from_file(model_dat, last=100):
    """np.mean / np.std over the last `last` rows of a model.dat written by HMCSample."""
    rows = np.loadtxt(model_dat, ndmin=2)
    rows = rows[-last:]
    return rows.mean(axis=0), rows.std(axis=0), rows.shape[0]


def summarize(model, dobs, rho_true=None):
    """Mean / std model of the device-side window, their forward responses and RMSD / RMSM.

    `model` is a GravMagModule whose chain has been run with a posterior window; the forward
    responses use the resident (weighted) kernel: d(m) = Aw (Wm m)."""
    eng = model._engine
    st = eng.posterior_read()
    wm = model.Wm.diagonal()
    dpre_mean = eng.forward(wm * st["mean"])
    dpre_std = eng.forward(wm * st["std"])
    out = {"n": st["n"], "mean": st["mean"], "std": st["std"], "dpre_mean": dpre_mean,
           "dpre_std": dpre_std,
           "RMSD": float(np.sqrt(np.linalg.norm(np.asarray(dobs) - dpre_mean) ** 2 / len(dobs)))}
    if rho_true is not None:
        out["RMSM"] = float(np.sqrt(np.linalg.norm(np.asarray(rho_true) - st["mean"]) ** 2 /
                                    st["mean"].shape[0]))
    return out
