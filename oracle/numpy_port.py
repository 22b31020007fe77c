"""NumPy/BLAS mirror of the reference's CPU hot path (the way the reference itself runs it).

TEST INFRASTRUCTURE ONLY (bench.py's cpu_baseline leg, tests/).  The reference evaluates the
potential with `np.dot(self.Aw, mw)` / `np.dot(self.Aw.T, r)` on a Fortran-ordered Aw
(inversion/potential.py:698,708 -> OpenBLAS dgemv, multi-threaded) and does the leapfrog
vector updates in NumPy (inversion/hmc.py:85-177).  This module restates exactly that, so the
baseline timed on the GPU box's host cores is the reference's formulation, not a slower
hand-rolled loop.  Damping / MS only (the regularisers of BASELINE.json's dense configs).
Parity: checked against oracle.Problem (itself pinned to the reference) in tests/.
"""
import numpy as np


class NumpyProblem(object):
    def __init__(self, Aw, dobs, mwapr, regularization="Damping", alpha=1.0, beta=0.01, wm=None,
                 grav_fix=None):
        self.Aw = np.asfortranarray(Aw)
        self.AwT = self.Aw.T                       # C-contiguous view, dgemv-T like the reference
        self.dobs_c = dobs - np.mean(dobs)
        self.mwapr, self.reg, self.alpha, self.beta = mwapr, regularization, alpha, beta
        self.wm2 = wm * wm if wm is not None else None
        self.grav_fix = grav_fix

    def misfit_and_grad(self, mw):
        dpre = np.dot(self.Aw, mw)                                   # potential.py:698
        dinv = dpre + self.grav_fix if self.grav_fix is not None else dpre
        r = (dinv - np.mean(dinv)) - self.dobs_c
        data_value = np.linalg.norm(r) ** 2                          # :706
        data_gradient = 2 * np.dot(self.AwT, r)                      # :708
        v = mw - self.mwapr
        if self.reg == "Damping":                                    # :775-784
            model_value, model_gradient = np.dot(v, v), 2 * v
        elif self.reg == "MS":                                       # :719-736
            v2 = v ** 2
            model_value = np.sum(self.wm2 * v2 / (v2 + self.beta))
            model_gradient = 2 * self.beta * self.wm2 * v / (v2 + self.beta) ** 2
        else:
            raise ValueError("numpy_port covers Damping and MS")
        return (data_value + self.alpha * model_value, data_gradient + self.alpha * model_gradient,
                dpre, data_value, model_value)

    def leapfrog(self, xcur, p0, dt, L, low, high, u):
        """hmc.py:85-177 ('mandatory' constraint)."""
        pnew, xnew = p0 * 1.0, xcur * 1.0
        K = np.dot(pnew, pnew) * 0.5
        U, grad, dsyn, U_data, U_model = self.misfit_and_grad(xnew)
        Hcur = K + U
        pnew -= dt * grad * 0.5
        for i in range(L):
            xnew += dt * pnew
            idx1, idx2 = xnew > high, xnew < low
            xnew[idx1], pnew[idx1] = high[idx1], -pnew[idx1]
            xnew[idx2], pnew[idx2] = low[idx2], -pnew[idx2]
            Unew, grad, dsyn_new, Unew_data, Unew_model = self.misfit_and_grad(xnew)
            pnew -= dt * grad if i < L - 1 else dt * grad * 0.5
        Hnew = np.dot(pnew, pnew) * 0.5 + Unew
        if Hnew < Hcur or u < np.exp(-(Hnew - Hcur)):
            return xnew, True, np.array([Unew, Unew_data, Unew_model, Hcur, Hnew]), dsyn_new
        return xcur, False, np.array([U, U_data, U_model, Hcur, Hnew]), dsyn
