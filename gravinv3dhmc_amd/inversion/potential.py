"""Potential energy U(x) and its gradient for HMC gravity inversion, evaluated on the GPU.

Host-side mirror of the reference's `inversion.potential.GravMagModule`
(inversion/potential.py:34-845): same constructor arguments, same public attributes the
drivers and the sampler read (`mshape`, `mask`, `mxs/mys/mzs`, `Wm`, `WmInv`, `WmSquare`,
`Aw`, `mesh`), same `kernelw()` / `misfit_and_grad()` signatures and error behaviour.
What differs is where the work happens: the kernel matrix is assembled, weighted and kept in
HBM by libgravhmc (one HIP launch each instead of the Python cell loop of prism.py:291-316
and the N x M Python loop of potential.py:241-244); `Aw` is a device handle that is copied
back only on request.
"""
import time

import numpy as np
from scipy.sparse import coo_matrix

from .. import _lib, mesher
from ..engine import DeviceMatrix, Engine


def _digest(a):
    """64-bit content digest of a contiguous float64 vector (for GravMagModule._use_reg)."""
    a = np.ascontiguousarray(a)
    try:
        import xxhash
        return xxhash.xxh3_64_intdigest(a.data)
    except ImportError:
        import zlib
        b = a.view(np.uint8)
        return (zlib.crc32(b) << 32) | zlib.adler32(b)


def _diag(values):
    row = np.arange(0, values.shape[0])
    return coo_matrix((values, (row, row))).tocsr()


class GravMagModule(object):
    """Gravity inversion model: mesh + sensitivity matrix + potential, on one MI355X.

    Parameters are the reference's (potential.py:35-58):

    * cartesian: mrange = (xmin, xmax, ymin, ymax, zmin, zmax), mspacing = (dz, dy, dx),
      obsurface = [xobs, yobs, height]; y East, x North, z Down.
    * spherical: mrange = (west, east, south, north, top, bottom), mspacing = (dr, dlat, dlon),
      obsurface = [lons, lats, height].
    * fixed / grav_fix: field of cells that do not take part in the inversion.
    * mratio: geometric growth of dz; mseg / mdivisionsection: piecewise dz.
    * weightfactor: exponent of the column-norm sensitivity weighting (0.5 = 2-norm).
    * wavelet: False, '1D' or '3D' (compressed forward operator).
    * mtopo=(x, y, topography) keyword: carve the mesh with a topography surface.
    * device: GPU ordinal (extension; the reference has no such argument).
    * shard: a `dist.Ranks` object: the cells of ONE model are split in column blocks over the
      ranks' GPUs (each holds N x M/world of G); shard_backend "rccl" or "gloo"; shard_planes=True
      splits in whole z-planes, which the Smoothness/TV regularisers need on a sharded model.
      shard_axis="rows": the OBSERVATIONS are split instead (row blocks, the model replicated: a gradient
      all-reduce per step); the wavelet-compressed forward works on row blocks only.
    * matrix_free: never store G; re-evaluate the prism / tesseroid entries in every potential
      evaluation (for kernels larger than HBM; the global tesseroid example: ~5x slower per step than
      the dense path).
    * shift_invariant: spherical models whose cell rows cover the full circle of longitudes with the
      observations on the same spacing (example/global/main_global.py:25-28): keep the table
      K[i, (c, k)] = T[c][class_i][(m_i - k) mod n] (35 MB for the global example) instead of G
      (4.25 GB); NotImplementedError from the constructor if the geometry lacks the structure.
    """

    def __init__(self, dobs, mrange, mspacing, obsurface, fixed=False, grav_fix=[],
                 mratio=1, mseg=False, mdivisionsection=[], weightfactor=0.5,
                 coordinate="cartesian", njobs=1, field="gravity",
                 mangle=(90, 0), wavelet=False, device=0, verbose=True, shard=None,
                 shard_backend="rccl", matrix_free=False, shard_planes=False, shift_invariant=False, shard_axis="cells",
                 **kwargs):
        self.dobs = dobs
        self.fixed = fixed
        self.grav_fix = grav_fix
        self.mrange = mrange
        self.mspacing = mspacing
        self.mratio = mratio
        self.weightfactor = weightfactor
        self.mseg = mseg
        self.mdivisionsection = mdivisionsection
        self.lonobs = obsurface[0]
        self.latobs = obsurface[1]
        self.heightobs = obsurface[2]
        self.inc, self.dec = mangle[0], mangle[1]
        self.njobs = njobs
        self.topocarve = False
        self.wavelet = wavelet
        self.device = device
        self._say = print if verbose else (lambda *a, **k: None)

        if field != "gravity" or coordinate not in ("cartesian", "spherical"):
            if field == "magnetic" and coordinate in ("cartesian", "spherical"):
                raise NotImplementedError(
                    "magnetic kernels are outside the accelerated hot path (gravity gz only)")
            raise ValueError("Please choose coordinate from(cartesian, spherical) and field "
                             "from(gravity, magnetic)!")
        if wavelet not in (False, None, '1D', '3D'):
            raise ValueError("wavelet must be False, '1D' or '3D'")
        self._say("Calculating {} field in {} coordinate.".format(field, coordinate))
        spherical = coordinate == "spherical"
        if spherical:
            mesh = (mesher.TesseroidMeshSegment(mrange, mspacing, mdivisionsection) if mseg
                    else mesher.TesseroidMesh(mrange, mspacing, mratio))
        else:
            mesh = (mesher.PrismMeshSegment(mrange, mspacing, mdivisionsection) if mseg
                    else mesher.PrismMesh(mrange, mspacing, mratio))
        for _key, value in kwargs.items():  # mtopo=(x, y, topography)  (potential.py:92-96)
            self.topocarve = True
            self.mask = mesh.carvetopo(value[0], value[1], value[2])
        mesh.addprop('density', np.zeros(mesh.size))
        self.mesh = mesh

        bounds = mesh.cell_bounds(active_only=True)
        N = int(np.asarray(self.lonobs).size)
        self._say("Start of calculate kernel")
        start = time.time()
        if shard is not None and shard.world > 1:
            if shard_axis not in ("cells", "rows"):
                raise ValueError("shard_axis must be 'cells' (column blocks) or 'rows' (row blocks)")
            if wavelet and shard_axis != "rows":
                raise NotImplementedError("wavelet forward on a kernel sharded in column blocks is not supported "
                                          "(shard_axis='rows': every rank compresses its own rows)")
            from ..dist import make_sharded_engine
            align = 1
            if shard_planes:
                if bounds.shape[0] != mesh.size:
                    raise ValueError("shard_planes needs the full (uncarved) mesh")
                align = int(mesh.shape[1]) * int(mesh.shape[2])
            eng = make_sharded_engine(N, bounds.shape[0], shard, backend=shard_backend, align=align, axis=shard_axis)
        else:
            eng = Engine(N, bounds.shape[0], device=device)
        if matrix_free:
            eng.set_matrix_free(True)      # (with wavelet: the compressor's rows are evaluated, never stored)
        if shift_invariant:
            if not spherical:
                raise NotImplementedError("the shift-invariant store is for spherical (tesseroid) models")
            eng.set_shift_invariant(True)
        self.matrix_free = bool(matrix_free or shift_invariant)
        eng.set_obs(self.lonobs, self.latobs, self.heightobs)
        if spherical:
            self._say("Number of effective tesseroids", bounds.shape[0])
            eng.set_cells(bounds, _lib.CELL_TESSEROID, 1.6)
        else:
            eng.set_cells(bounds, _lib.CELL_PRISM)
        eng.build_G()
        if spherical and eng.kernel_stats()["warn_cells"] > 0:
            import warnings
            from ..gravmag.tesseroid import _WARN_DIVIDE
            warnings.warn(_WARN_DIVIDE, RuntimeWarning)
        if not spherical:
            self._say("kernel.shape", (N, bounds.shape[0]))
        self._say("End of calculate kernel:%.6f s" % (time.time() - start))
        self._engine = eng

        self.mshape = mesh.shape
        self.mxs, self.mys, self.mzs = mesh.get_xs(), mesh.get_ys(), mesh.get_zs()
        self._say("Start to weight kernel")
        start = time.time()
        self.sensitivityWeighting()
        self._say("End of weighting kernel: %.6f s" % (time.time() - start))
        eng.set_data(self.dobs, self.grav_fix if self.fixed else None)
        if wavelet in ('1D', '3D'):
            # gravmag/compressor1D.py / compressor3D.py: db4, level 2, 'periodization',
            # threshold 1e-3, CSR -- built and applied on the device
            self._say("Using {} wavelet to compress kernel.".format(wavelet))
            eng.compress_wavelet(3 if wavelet == '3D' else 1, self.mshape, 0.001, 2)

    @property
    def Awcp(self):
        """Compressed kernel as scipy CSR (copied from the device on request)."""
        if not self.wavelet:
            raise AttributeError("Awcp exists only with wavelet='1D' or '3D'")
        return self._engine.download_csr()

    # ------------------------------------------------------------------ weighting
    def sensitivityWeighting(self):
        """Column-norm weighting Wm and Aw = A Wm^-1 (potential.py:232-264), on the device."""
        wm = self._engine.weight(self.weightfactor)
        with np.errstate(divide='ignore'):
            inv = 1.0 / wm
        self.Wm = _diag(wm)
        self.WmInv = _diag(inv)
        self.WmSquare = _diag(wm * wm)
        self.Aw = DeviceMatrix(self._engine)

    def kernelw(self):
        """(Aw, WmInv, Wm) as the sampler expects (potential.py:584-589); Aw is a device handle."""
        return self.Aw, self.WmInv, self.Wm

    # ------------------------------------------------------------------ potential
    def _to_mw(self, x, low, high, constraint, log_fator):
        if constraint == 'logarithmic':
            return (low + high * np.e ** (log_fator * x)) / (1 + np.e ** (log_fator * x))
        elif constraint == 'mandatory':
            return x
        raise ValueError("Please choose right boundary constraint(mandatory, logarithmic)!")

    def _use_reg(self, regulization, alpha, beta, mwapr):
        if regulization not in _lib.REG_KINDS:
            raise ValueError("Please choose regularization from 'MS','Damping', 'Smoothness', 'TV'.")
        mwapr = np.asarray(mwapr, dtype=np.float64)
        # The decision to (re)send the regulariser must be the same on every rank of a sharded
        # model (gh_set_reg is collective there) and must notice in-place edits: it depends only
        # on the VALUES -- kind, alpha, beta and the content of the full mwapr vector -- never on
        # addresses.  The content is compared through a 64-bit digest of its bytes (xxh3: ~10 GB/s,
        # one pass, no second copy of the vector kept; zlib.crc32 pair where xxhash is missing).
        key = (regulization, float(alpha), float(beta))
        digest = (mwapr.shape, _digest(mwapr))
        last = self._engine._reg_key
        if last is None or last[0] != key or last[1] != digest:
            m_model = getattr(self._engine, "M_global", self._engine.M)
            if regulization in ("Smoothness", "TV") and int(np.prod(self.mshape)) != m_model:
                raise ValueError("Smoothness/TV need the full (uncarved) mesh: shape %r has %d "
                                 "cells, model has %d" % (self.mshape, int(np.prod(self.mshape)),
                                                          m_model))
            self._engine.set_reg(regulization, alpha, beta, self.mshape, mwapr)
            self._engine._reg_key = (key, digest)

    def misfit_and_grad(self, x, mwapr, low, high, constraint, log_fator, alpha,
                        regulization='Damping', beta=0.01):
        """(misfit, grad, dpre, data_value, model_value) -- potential.py:812-845."""
        mw = self._to_mw(x, low, high, constraint, log_fator)
        self._use_reg(regulization, alpha, beta, mwapr)
        return self._engine.misfit_and_grad(mw)

    # wrappers the reference keeps for an (unused) adaptive regularisation factor
    def data(self, x, low, high, constraint, log_fator):
        mw = self._to_mw(x, low, high, constraint, log_fator)
        self._use_reg("Damping", 0.0, 0.01, np.zeros(self._engine.M))
        return self._engine.misfit_and_grad(mw)[3]

    def _model(self, kind, x, mwapr, low, high, constraint, log_fator, beta=0.01):
        mw = self._to_mw(x, low, high, constraint, log_fator)
        self._use_reg(kind, 1.0, beta, mwapr)
        return self._engine.misfit_and_grad(mw)[4]

    def model_MS(self, x, mwapr, low, high, constraint, log_fator, beta):
        return self._model("MS", x, mwapr, low, high, constraint, log_fator, beta)

    def model_Damping(self, x, mwapr, low, high, constraint, log_fator):
        return self._model("Damping", x, mwapr, low, high, constraint, log_fator)

    def model_Smoothness(self, x, mwapr, low, high, constraint, log_fator):
        return self._model("Smoothness", x, mwapr, low, high, constraint, log_fator)

    def model_TV(self, x, mwapr, low, high, constraint, log_fator, beta):
        return self._model("TV", x, mwapr, low, high, constraint, log_fator, beta)
