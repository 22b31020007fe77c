"""db4 / periodization discrete wavelet transform, restated from PyWavelets' published algorithm.

TEST INFRASTRUCTURE ONLY.  The reference calls the third-party package PyWavelets (`import pywt`,
NOT vendored under /root/reference and not pinned to a version anywhere in the reference:
gravmag/compressor1D.py:23,32-33,54-55, gravmag/compressor3D.py:23,34-35,60-61) with
`pywt.Wavelet('db4')`, `wavedec`/`wavedecn(mode='periodization', level=2)` and
`pywt.coeffs_to_array`.  pywt is not installed in the build image, so this restatement is pinned
ONLY by the misfit lines of the reference's committed wavelet runs
(example/uniformgrid/logout_T1.txt:33-53, example/segmentgrid/logout_T0.txt:41-57): 7 printed
digits.  Beyond those digits the sub-path is "parity unpinned" (DESIGN.md section 2).

Algorithm (PyWavelets `downsampling_convolution_periodization`, step 2): a signal of odd length
n is first extended by repeating its last sample (n' = n + n%2), then treated as n'-periodic;
    out[o] = sum_{j=0}^{F-1} filt[j] * x[(F/2 + 2 o - j) mod n'],   o = 0 .. n'/2 - 1
with filt = dec_lo for the approximation and dec_hi for the detail (F = 8 for db4).
"""
import numpy as np

# db4 analysis low-pass filter (Daubechies, 8 taps) in PyWavelets' dec_lo order; these digits
# are orthonormal to 1e-17 (sum h^2 = 1, sum h = sqrt 2).  dec_hi is the quadrature mirror:
# dec_hi[j] = (-1)^(j+1) dec_lo[7-j].
DEC_LO = np.array([-0.010597401785069032, 0.0328830116668852, 0.030841381835560764,
                   -0.18703481171909309, -0.027983769416859854, 0.6308807679298589,
                   0.7148465705529157, 0.2303778133088965])
DEC_HI = np.array([-0.2303778133088965, 0.7148465705529157, -0.6308807679298589,
                   -0.027983769416859854, 0.18703481171909309, 0.030841381835560764,
                   -0.0328830116668852, -0.010597401785069032])
LEVELS = 2           # compressor*.py: Nlevel = 2
THRESHOLD = 0.001    # compressor*.py: thrg


def dwt_axis(x, axis):
    """One periodized analysis step along `axis`: returns (approx, detail), length ceil(n/2)."""
    x = np.moveaxis(np.asarray(x, dtype=np.float64), axis, -1)
    n = x.shape[-1]
    if n % 2:
        x = np.concatenate([x, x[..., -1:]], axis=-1)
        n += 1
    half = n // 2
    o = np.arange(half)
    a = np.zeros(x.shape[:-1] + (half,))
    d = np.zeros_like(a)
    for j in range(8):
        idx = (4 + 2 * o - j) % n
        xs = x[..., idx]
        a += DEC_LO[j] * xs
        d += DEC_HI[j] * xs
    return np.moveaxis(a, -1, axis), np.moveaxis(d, -1, axis)


def wavedec1_packed(x, levels=LEVELS):
    """pywt.coeffs_to_array(pywt.wavedec(x, 'db4', 'periodization', level))[0] along the last axis:
    [cA_L | cD_L | ... | cD_1]."""
    a = np.asarray(x, dtype=np.float64)
    details = []
    for _ in range(levels):
        a, d = dwt_axis(a, -1)
        details.append(d)
    return np.concatenate([a] + details[::-1], axis=-1)


def wavedec3_packed(x, shape, levels=LEVELS):
    """pywt.coeffs_to_array(pywt.wavedecn(x.reshape(shape), 'db4', 'periodization', level))[0],
    flattened; leading axes of x (e.g. kernel rows) are carried along.  Gaps left by odd
    lengths are zero-filled exactly as coeffs_to_array(padding=0) does."""
    x = np.asarray(x, dtype=np.float64)
    lead = x.shape[:-1]
    a = x.reshape(lead + tuple(shape))
    nl = len(lead)
    levels_out = []
    for _ in range(levels):
        bands = {"": a}
        for ax in range(3):
            nb = {}
            for key, v in bands.items():
                lo, hi = dwt_axis(v, nl + ax)
                nb[key + "a"], nb[key + "d"] = lo, hi
            bands = nb
        a = bands.pop("aaa")
        levels_out.append(bands)
    # pack: approximation at the origin, every detail level around the running block
    a_shape = list(a.shape[nl:])
    blocks = [((0, 0, 0), a)]
    for bands in levels_out[::-1]:
        d_shape = list(bands["ddd"].shape[nl:])
        for key, v in bands.items():
            off = tuple(a_shape[i] if key[i] == "d" else 0 for i in range(3))
            blocks.append((off, v))
        a_shape = [a_shape[i] + d_shape[i] for i in range(3)]
    out = np.zeros(lead + tuple(a_shape))
    for off, v in blocks:
        s = v.shape[nl:]
        out[(Ellipsis,) + tuple(slice(off[i], off[i] + s[i]) for i in range(3))] = v
    return out.reshape(lead + (-1,)), tuple(a_shape)


def compress_kernel(Aw, dims, shape=None, thr=THRESHOLD):
    """gravmag/compressor3D.py:17-44 / compressor1D.py:17-42: row-wise DWT of the weighted
    kernel, hard threshold |c| < thr -> 0, CSR."""
    from scipy.sparse import csr_matrix
    Aw = np.ascontiguousarray(Aw)
    if dims == 3:
        C, _ = wavedec3_packed(Aw, shape)
    else:
        C = wavedec1_packed(Aw)
    C[np.abs(C) < thr] = 0
    return csr_matrix(C)


def model_coeffs(mw, dims, shape=None):
    """compressor3D.py:47-63 / compressor1D.py:45-56: coefficients of the model vector."""
    if dims == 3:
        return wavedec3_packed(mw, shape)[0]
    return wavedec1_packed(mw)
