"""NumPy restatement of the reference's conjugate-gradient optimiser (test infrastructure only).

inversion/reginv.py:120-149 (newkernel), :248-355 (data / model terms), :357-492 (CG), with the
reference's quirks: no mean removal in the data term, MS gradient denominator (mw^2 + beta)^2,
clamp of the unweighted model after every update.  Pinned by tests/golden/cg_small.npz, generated
from the reference class itself (oracle/make_golden.py::cg_small).
"""
import numpy as np

from oracle import oracle


def cg(K, dobs, shape, initialModel, apriorModel, boundary, regularization="MS", beta=0.01, q=0.9,
       maxk=100):
    Aw, wm = oracle.col_weight(K)
    N, M = Aw.shape
    wm2 = wm * wm
    R = oracle.fd3d_dense(shape) if regularization in ("Smoothness", "TV") else None

    def data(mw):
        return np.linalg.norm(Aw @ mw - dobs) ** 2

    def data_g(mw):
        return 2 * Aw.T @ (Aw @ mw - dobs)

    def model(mw):
        v = mw - mwapr
        if regularization == "MS":
            return np.sum(wm2 * v ** 2 / (v ** 2 + beta))
        if regularization == "Damping":
            return v @ v
        t = R @ v
        return t @ t if regularization == "Smoothness" else np.sum(np.sqrt(t ** 2 + beta))

    def model_g(mw):
        v = mw - mwapr
        if regularization == "MS":
            return 2 * beta * wm2 * v / (mw * mw + beta) ** 2
        if regularization == "Damping":
            return 2 * v
        t = R @ v
        return 2 * R.T @ t if regularization == "Smoothness" else R.T @ (t / np.sqrt(t ** 2 + beta))

    def step(mw, I, Iw, alpha):
        kstep = (Iw @ I) / (np.linalg.norm(Aw @ Iw) ** 2 + alpha * np.linalg.norm(Iw) ** 2)
        m = (mw - kstep * Iw) / wm
        m[m < boundary[0]] = boundary[0]
        m[m > boundary[1]] = boundary[1]
        return wm * m

    mw, mwapr = wm * initialModel, wm * apriorModel
    dm, mm, rf = [], [], []
    mw_new = mw
    for k in range(maxk):
        if k == 0:
            alpha = 0
        elif k == 1:
            alpha = data(mw_new) / model(mw_new)
        elif data(mw) - data(mw_new) < 0.01 * data(mw):
            alpha = q * alpha
        rf.append(alpha)
        if k == 0:
            dm.append(data(mw) / N)
            I = data_g(mw) + alpha * model_g(mw)
            mm.append(model(mw) / M)
            Iw = I
            mw_new = step(mw, I, Iw, alpha)
        else:
            I_old, Iw_old = I, Iw
            mw = mw_new
            I = data_g(mw) + alpha * model_g(mw)
            Iw = I + (np.linalg.norm(I) ** 2 / np.linalg.norm(I_old) ** 2) * Iw_old
            mw_new = step(mw, I, Iw, alpha)
            dm.append(data(mw_new) / N)
            mm.append(model(mw_new) / M)
            if dm[-1] < 0.001:
                break
    return mw_new / wm, Aw @ mw_new, np.array(dm), np.array(mm), np.array(rf, dtype=float)


def bootstrap(K, dobs, boundary, initialModel, samples=3, beta=0.01, maxk=5, wavelet=None, shape=None):
    """inversion/reginv.py:494-755 (BootStrap.BSCG / .CG): CG with the MS variant of that class
    (no prior, beta squared: :599-629) on row-resampled data, seed = sample index.
    wavelet = '1D' / '3D' (reginv.py:546-553): the predicted data of the data term and of its gradient come
    from the compressed UNRESAMPLED kernel, `modelcompressor(mw, self.Awcp)` (:590-593, :608-617), i.e. in the
    ORIGINAL row order, and are compared with the RESAMPLED observations -- the reference's behaviour as
    written (the step length still uses the resampled dense kernel, :655).  PyWavelets is not installed here:
    the transform is oracle/wavelet.py's restatement (pinned by the reference's wavelet logs only)."""
    Aw, wm = oracle.col_weight(K)
    N, M = Aw.shape
    wm2 = wm * wm
    b2 = beta ** 2
    pred = None
    if wavelet:
        from . import wavelet as ow
        dims = 3 if wavelet == "3D" else 1
        Awcp = ow.compress_kernel(Aw, dims, shape)
        pred = lambda mw: Awcp @ ow.model_coeffs(mw, dims, shape)

    def run(A, d):
        fwd = pred if pred is not None else (lambda mw: A @ mw)
        data = lambda mw: np.linalg.norm(fwd(mw) - d) ** 2
        data_g = lambda mw: 2 * A.T @ (fwd(mw) - d)
        model = lambda mw: np.sum(wm2 * mw * mw / (mw * mw + b2))
        model_g = lambda mw: 2 * wm2 * (mw * b2) / (mw * mw + b2) ** 2

        def step(mw, I, Iw, alpha):
            kstep = (Iw @ I) / (np.linalg.norm(A @ Iw) ** 2 + alpha * np.linalg.norm(Iw) ** 2)
            m = (mw - kstep * Iw) / wm
            m[m < boundary[0]] = boundary[0]
            m[m > boundary[1]] = boundary[1]
            return wm * m

        mw = wm * initialModel
        dm, mm, rf = [], [], []
        for k in range(maxk):
            if k == 0:
                alpha = 0
            elif k == 1:
                alpha = data(mw_new) / model(mw_new)
            elif data(mw) - data(mw_new) < 0.01 * data(mw):
                alpha = 0.9 * alpha
            rf.append(alpha)
            if k == 0:
                I = data_g(mw) + alpha * model_g(mw)
                Iw = I
                mw_new = step(mw, I, Iw, alpha)
            else:
                I_old, Iw_old = I, Iw
                mw = mw_new
                I = data_g(mw) + alpha * model_g(mw)
                Iw = I + (np.linalg.norm(I) ** 2 / np.linalg.norm(I_old) ** 2) * Iw_old
                mw_new = step(mw, I, Iw, alpha)
                if data(mw_new) < 0.1:
                    break
                dm.append(data(mw_new) / N)
                mm.append(model(mw_new) / M)
        return mw_new / wm, dm, mm, rf

    models = np.zeros((samples, M))
    dms, mms, rfs = np.zeros((samples, maxk - 1)), np.zeros((samples, maxk - 1)), np.zeros((samples, maxk))
    for s in range(samples):
        np.random.seed(s)
        idx = np.random.choice(np.arange(N), size=N, replace=True, p=None)
        models[s], dms[s], mms[s], rfs[s] = run(Aw[idx, :], dobs[idx])
    return models, dms, mms, rfs
