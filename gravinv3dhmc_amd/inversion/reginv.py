"""Deterministic regularised inversion (conjugate gradient) on the device kernels.

Host-side mirror of the reference's `inversion.reginv.ConjugateGradient`
(inversion/reginv.py:22-492): same constructor arguments, same `CG(...)` signature, return
values and console lines.  It is the second caller of the hot path's primitives:
every `Aw @ v` / `Aw.T @ r` is a sweep of the resident kernel matrix (gh_forward / gh_adjoint),
the regulariser terms come from gh_reg_eval; only O(M) vector arithmetic of the CG recurrence
stays in NumPy.  Quirks of the reference kept: no mean removal in the data term
(reginv.py:248-269), MS gradient denominator (mw^2 + beta)^2 (reginv.py:288-292), density clamp in
unweighted space after every update (:434-438, :464-468).
"""
import time

import numpy as np
from scipy.sparse import coo_matrix

from .. import _lib, mesher
from ..engine import DeviceMatrix, Engine


def _diag(values):
    row = np.arange(0, values.shape[0])
    return coo_matrix((values, (row, row))).tocsr()


class ConjugateGradient(object):
    def __init__(self, dobs, mrange, mspacing, obsurface, mratio=1, njobs=1, coordinate="cartesian",
                 field="gravity", mangle=(90, 0), wavelet=False, device=0, verbose=True, **kwargs):
        self.dobs = np.asarray(dobs, dtype=np.float64)
        self.mrange, self.mspacing, self.mratio = mrange, mspacing, mratio
        self.lonobs, self.latobs, self.heightobs = obsurface[0], obsurface[1], obsurface[2]
        self.njobs = njobs
        self.inc, self.dec = mangle[0], mangle[1]
        self.wavelet = wavelet
        say = print if verbose else (lambda *a, **k: None)
        if field != "gravity" or coordinate not in ("cartesian", "spherical"):
            if field == "magnetic" and coordinate in ("cartesian", "spherical"):
                raise NotImplementedError("magnetic kernels are outside the accelerated hot path")
            raise ValueError("Please choose coordinate from(cartesian, spherical) and field "
                             "from(gravity, magnetic)!")
        say("Calculating {} field in {} coordinate.".format(field, coordinate))
        spherical = coordinate == "spherical"
        mesh = (mesher.TesseroidMesh if spherical else mesher.PrismMesh)(mrange, mspacing, mratio)
        for _key, value in kwargs.items():
            self.topocarve = True
            self.mask = mesh.carvetopo(value[0], value[1], value[2])
        mesh.addprop('density', np.zeros(mesh.size))
        self.mesh = mesh
        bounds = mesh.cell_bounds(active_only=True)
        say("Start of calculate kernel")
        start = time.time()
        eng = Engine(int(np.asarray(self.lonobs).size), bounds.shape[0], device=device)
        eng.set_obs(self.lonobs, self.latobs, self.heightobs)
        eng.set_cells(bounds, _lib.CELL_TESSEROID if spherical else _lib.CELL_PRISM, 1.6)
        eng.build_G()
        say("End of calculate kernel:%.6f s" % (time.time() - start))
        self._engine = eng
        self.mshape = mesh.shape
        self.dsize, self.msize = eng.N, eng.M
        self.mxs, self.mys, self.mzs = mesh.get_xs(), mesh.get_ys(), mesh.get_zs()
        self.newkernel()
        if wavelet in ('1D', '3D'):
            say("Using {} wavelet to compress kernel.".format(wavelet))
            eng.compress_wavelet(3 if wavelet == '3D' else 1, self.mshape, 0.001, 2)

    def newkernel(self):
        """Column-norm weighting (reginv.py:120-149), on the device."""
        wm = self._engine.weight(0.5)
        with np.errstate(divide='ignore'):
            inv = 1.0 / wm
        self.Wm, self.WmInv, self.WmSquare = _diag(wm), _diag(inv), _diag(wm * wm)
        self.Aw = DeviceMatrix(self._engine)

    # ---- terms of the objective (reginv.py:248-355) ------------------------------------
    def _dpre(self, mw):
        return self._engine.forward_wavelet(mw) if self.wavelet else self._engine.forward(mw)

    def data(self, mw):
        return np.linalg.norm(self._dpre(mw) - self.dobs) ** 2

    def data_gfun(self, mw):
        return 2 * self._engine.adjoint(self._dpre(mw) - self.dobs)

    def _reg(self, kind, mw, mwapr, beta, grad):
        v, g = self._engine.reg_eval(kind, mw, mwapr, beta, self.mshape, ms_grad_den_mw=True,
                                     want_grad=grad)
        return g if grad else v

    def model_MS(self, mw, mwapr, beta):
        return self._reg("MS", mw, mwapr, beta, False)

    def model_gfun_MS(self, mw, mwapr, beta):
        return self._reg("MS", mw, mwapr, beta, True)

    def model_Damping(self, mw, mwapr):
        return self._reg("Damping", mw, mwapr, 0.01, False)

    def model_gfun_Damping(self, mw, mwapr):
        return self._reg("Damping", mw, mwapr, 0.01, True)

    def model_Smoothness(self, mw, mwapr):
        return self._reg("Smoothness", mw, mwapr, 0.01, False)

    def model_gfun_Smoothness(self, mw, mwapr):
        return self._reg("Smoothness", mw, mwapr, 0.01, True)

    def model_TV(self, mw, mwapr, beta):
        return self._reg("TV", mw, mwapr, beta, False)

    def model_gfun_TV(self, mw, mwapr, beta):
        return self._reg("TV", mw, mwapr, beta, True)

    # ---- the iteration (reginv.py:357-492) -----------------------------------------------
    def CG(self, initialModel, apriorModel, boundary, regularization='MS', beta=0.01, q=0.9, maxk=100):
        """Returns model_inv, data_inv, data_misfit, model_misfit, regul_factor."""
        if regularization not in ("MS", "Damping", "Smoothness", "TV"):
            raise ValueError("Please choose regularization from 'MS','Damping', 'Smoothness', 'TV'.")

        def model(mw_):
            if regularization == "MS":
                return self.model_MS(mw_, mwapr, beta)
            if regularization == "Damping":
                return self.model_Damping(mw_, mwapr)
            if regularization == "Smoothness":
                return self.model_Smoothness(mw_, mwapr)
            return self.model_TV(mw_, mwapr, beta)

        def model_g(mw_):
            if regularization == "MS":
                return self.model_gfun_MS(mw_, mwapr, beta)
            if regularization == "Damping":
                return self.model_gfun_Damping(mw_, mwapr)
            if regularization == "Smoothness":
                return self.model_gfun_Smoothness(mw_, mwapr)
            return self.model_gfun_TV(mw_, mwapr, beta)

        def step(mw_, I_, Iw_, alpha_):
            kstep = np.dot(Iw_.T, I_) / (np.linalg.norm(self._engine.forward(Iw_)) ** 2 +
                                         alpha_ * np.linalg.norm(Iw_) ** 2)
            new = mw_ - kstep * Iw_
            mtemp = self.WmInv @ new
            mtemp[mtemp < rhomin] = rhomin
            mtemp[mtemp > rhomax] = rhomax
            return self.Wm @ mtemp

        mw = self.Wm @ initialModel
        mwapr = self.Wm @ apriorModel
        rhomin, rhomax = boundary[0], boundary[1]
        data_misfit, model_misfit, regul_factor = [], [], []
        mw_new = mw
        for k in range(0, maxk):
            print("CG iteration: ", k + 1)
            if k == 0:
                alpha = 0
            elif k == 1:
                alpha = self.data(mw_new) / model(mw_new)
            else:
                d_old = self.data(mw)
                if d_old - self.data(mw_new) < 0.01 * d_old:
                    alpha = q * alpha
            regul_factor.append(alpha)
            if k == 0:
                data_misfit.append(self.data(mw) / self.dsize)
                I = self.data_gfun(mw) + alpha * model_g(mw)
                model_misfit.append(model(mw) / self.msize)
                Iw = I
                mw_new = step(mw, I, Iw, alpha)
            if k > 0:
                I_old, Iw_old = I, Iw
                mw = mw_new
                I = self.data_gfun(mw) + alpha * model_g(mw)
                mu = np.linalg.norm(I) ** 2 / np.linalg.norm(I_old) ** 2
                Iw = I + mu * Iw_old
                mw_new = step(mw, I, Iw, alpha)
                d_new = self.data(mw_new) / self.dsize
                data_misfit.append(d_new)
                print("Normed data error:", d_new)
                m_new = model(mw_new) / self.msize
                model_misfit.append(m_new)
                print("Normed model error:", m_new)
                if d_new < 0.001:
                    print("Normed data error is {} < 0.001, stop iteration!".format(d_new))
                    break
        model_inv = self.WmInv @ mw_new
        data_inv = self._engine.forward(mw_new)   # = A @ model_inv (reginv.py:490)
        return model_inv, data_inv, data_misfit, model_misfit, regul_factor


class BootStrap(object):
    """Bootstrap of the observations around the CG inversion (mirror of reginv.py:494-755).

    The reference builds, for every replicate, a row-resampled copy of the weighted kernel
    (`AwSample[i, :] = Aw[indexSample[i], :]`, reginv.py:736-741).  Sampling rows with
    replacement is the same as weighting row r by the number of times it was drawn:
        |Aw_s v - d_s|^2 = sum_r c_r (Aw[r].v - d[r])^2,   Aw_s^T (Aw_s v - d_s) = Aw^T (c * (Aw v - d))
    so the resident kernel is used as it is (no second matrix, no gather), with the count vector
    c applied to the N-vector between the forward and the adjoint sweep.  MS stabiliser of that
    class: no prior, beta squared (reginv.py:599-629)."""

    def __init__(self, mrange, mspacing, obsurface, dobs, boundary, samples=100, beta=0.01, maxk=100,
                 mratio=1, njobs=1, wavelet=False, device=0, verbose=True, **kwargs):
        if wavelet not in (False, None, '1D', '3D'):
            raise ValueError("wavelet must be False, '1D' or '3D'")
        self.mrange, self.mspacing, self.mratio = mrange, mspacing, mratio
        self.lonobs, self.latobs, self.heightobs = obsurface[0], obsurface[1], obsurface[2]
        self.boundary, self.samples, self.njobs = boundary, samples, njobs
        self.dobs = np.asarray(dobs, dtype=np.float64)
        self.maxk, self.beta, self.wavelet = maxk, beta, wavelet
        say = print if verbose else (lambda *a, **k: None)
        say("Calculating gravity field using prism.")
        mesh = mesher.PrismMesh(mrange, mspacing, mratio)
        for _key, value in kwargs.items():
            self.mask = mesh.carvetopo(value[0], value[1], value[2])
        mesh.addprop('density', np.zeros(mesh.size))
        self.mesh = mesh
        bounds = mesh.cell_bounds(active_only=True)
        start = time.time()
        eng = Engine(int(np.asarray(self.lonobs).size), bounds.shape[0], device=device)
        eng.set_obs(self.lonobs, self.latobs, self.heightobs)
        eng.set_cells(bounds, _lib.CELL_PRISM)
        eng.build_G()
        say("End of calculate kernel:", time.time() - start)
        self._engine = eng
        self.mshape = mesh.shape
        self.dsize, self.msize = eng.N, eng.M
        self.mxs, self.mys, self.mzs = mesh.get_xs(), mesh.get_ys(), mesh.get_zs()
        wm = eng.weight(0.5)
        with np.errstate(divide='ignore'):
            inv = 1.0 / wm
        self.Wm, self.WmInv, self.WmSquare = _diag(wm), _diag(inv), _diag(wm * wm)
        self.Aw = DeviceMatrix(eng)
        self._zero = np.zeros(eng.M)
        self._index = None
        if wavelet in ('1D', '3D'):
            # gravmag/compressor1D.py / compressor3D.py on the weighted kernel (reginv.py:546-553)
            say("Using {} wavelet to compress kernel.".format(wavelet))
            eng.compress_wavelet(3 if wavelet == '3D' else 1, self.mshape, 0.001, 2)

    # terms of one replicate: `counts[r]` = how often observation r was drawn
    def data(self, mw, counts, dobs):
        if self.wavelet:
            # reginv.py:590-593: the compressed, UNRESAMPLED operator predicts the data in the original row
            # order; they are compared with the resampled observations as they are (the reference's own
            # behaviour, kept)
            res = self._engine.forward_wavelet(mw) - dobs[self._index]
            return float(np.sum(res * res))
        res = self._engine.forward(mw) - dobs
        return float(np.sum(counts * res * res))

    def data_gfun(self, mw, counts, dobs):
        if self.wavelet:
            # 2 AwSample^T (dpre - dobsSample) (reginv.py:608-617): AwSample[i] = Aw[index[i]], so the
            # residual of draw i goes to row index[i] of the one resident kernel
            res = self._engine.forward_wavelet(mw) - dobs[self._index]
            return 2 * self._engine.adjoint(np.bincount(self._index, weights=res, minlength=self.dsize))
        return 2 * self._engine.adjoint(counts * (self._engine.forward(mw) - dobs))

    def model_MS(self, mw):
        return self._engine.reg_eval("MS", mw, self._zero, self.beta ** 2, self.mshape, want_grad=False)[0]

    def model_gfun_MS(self, mw):
        return self._engine.reg_eval("MS", mw, self._zero, self.beta ** 2, self.mshape)[1]

    def CG(self, counts, dobs, initialModel):
        """One replicate (reginv.py:631-713); `counts` replaces the resampled kernel."""
        mw = self.Wm @ initialModel
        rhomin, rhomax = self.boundary[0], self.boundary[1]
        q = 0.9
        data_misfit, model_misfit, regul_factor = [], [], []

        def step(mw_, I_, Iw_, alpha_):
            fw = self._engine.forward(Iw_)
            kstep = np.dot(Iw_.T, I_) / (np.sum(counts * fw * fw) + alpha_ * np.linalg.norm(Iw_) ** 2)
            mtemp = self.WmInv @ (mw_ - kstep * Iw_)
            mtemp[mtemp < rhomin] = rhomin
            mtemp[mtemp > rhomax] = rhomax
            return self.Wm @ mtemp

        mw_new = mw
        for k in range(0, self.maxk):
            if k == 0:
                alpha = 0
            elif k == 1:
                alpha = self.data(mw_new, counts, dobs) / self.model_MS(mw_new)
            else:
                d_old = self.data(mw, counts, dobs)
                if d_old - self.data(mw_new, counts, dobs) < 0.01 * d_old:
                    alpha = q * alpha
            regul_factor.append(alpha)
            if k == 0:
                I = self.data_gfun(mw, counts, dobs) + alpha * self.model_gfun_MS(mw)
                Iw = I
                mw_new = step(mw, I, Iw, alpha)
            if k > 0:
                I_old, Iw_old = I, Iw
                mw = mw_new
                I = self.data_gfun(mw, counts, dobs) + alpha * self.model_gfun_MS(mw)
                mu = np.linalg.norm(I) ** 2 / np.linalg.norm(I_old) ** 2
                Iw = I + mu * Iw_old
                mw_new = step(mw, I, Iw, alpha)
                d_new = self.data(mw_new, counts, dobs)
                if d_new < 0.1:
                    print("Data error is {} < 0.1, stop iteration!".format(d_new))
                    break
                data_misfit.append(d_new / self.dsize)
                m_new = self.model_MS(mw_new) / self.msize
                model_misfit.append(m_new)
                print(d_new / self.dsize)
                print(m_new)
            print("CG iteration: ", k)
        return self.WmInv @ mw_new, data_misfit, model_misfit, regul_factor

    def BSCG(self, initialModel):
        """`samples` replicates, replicate s drawn with np.random.seed(s) (reginv.py:715-755)."""
        model_inv_all = np.zeros((self.samples, self.msize))
        data_misfit_all = np.zeros((self.samples, self.maxk - 1))
        model_misfit_all = np.zeros((self.samples, self.maxk - 1))
        regul_factor_all = np.zeros((self.samples, self.maxk))
        for sample in range(self.samples):
            print("*********Sample {}*********".format(sample + 1))
            np.random.seed(sample)
            index = np.arange(0, self.dsize)
            indexSample = np.random.choice(index, size=self.dsize, replace=True, p=None)
            counts = np.bincount(indexSample, minlength=self.dsize).astype(np.float64)
            self._index = indexSample
            model_inv, data_misfit, model_misfit, regul_factor = self.CG(counts, self.dobs, initialModel)
            model_inv_all[sample, :] = model_inv
            data_misfit_all[sample, :] = data_misfit
            model_misfit_all[sample, :] = model_misfit
            regul_factor_all[sample, :] = regul_factor
        return model_inv_all, data_misfit_all, model_misfit_all, regul_factor_all
