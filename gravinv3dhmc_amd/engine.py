"""numpy-facing wrapper of one libgravhmc context (one problem resident on one MI355X)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f64, ptr


class DeviceMatrix(object):
    """Handle of the (weighted) kernel matrix resident in HBM.

    Stands in for the dense ndarray the reference hands around as `Aw`
    (potential.py:584-589); `np.asarray(handle)` / `.to_numpy()` copies it back,
    Fortran-ordered N x M, only when somebody really asks for it."""

    def __init__(self, engine):
        self._engine = engine
        self.shape = (engine.N, engine.M)
        self.dtype = np.dtype(np.float64)
        self.ndim = 2

    def to_numpy(self):
        return self._engine.download_G()

    def __array__(self, dtype=None, copy=None):
        a = self.to_numpy()
        return a if dtype is None else a.astype(dtype)

    @property
    def T(self):
        return self.to_numpy().T

    def __repr__(self):
        return "DeviceMatrix(shape=%r, device=%d)" % (self.shape, self._engine.device)


class Engine(object):
    def __init__(self, N, M, device=0):
        self._lib = _lib.load()
        self.N, self.M, self.device = int(N), int(M), int(device)
        h = C.c_void_p()
        rc = self._lib.gh_create(C.byref(h), self.device, self.N, self.M)
        check(rc, None)
        self._h = h
        self._reg_key = None
        self._chain_valid = False

    # -- lifetime -----------------------------------------------------------------
    def close(self):
        self.__dict__.pop("_p0_pool", None)     # (the blocks go with the context)
        if getattr(self, "_h", None):
            self._lib.gh_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        check(rc, self._h)

    def _vecM(self, v, what):
        """Contiguous float64 copy/view of a model vector; the library copies exactly M doubles."""
        v = f64(v)
        if v.shape != (self.M,):
            raise ValueError("%s must have M = %d entries, got shape %r" % (what, self.M, v.shape))
        return v

    def _vecN(self, v, what):
        v = f64(v)
        if v.shape != (self.N,):
            raise ValueError("%s must have N = %d entries, got shape %r" % (what, self.N, v.shape))
        return v

    def device_info(self):
        name = C.create_string_buffer(256)
        cus, mem = C.c_int(0), C.c_int64(0)
        self._chk(self._lib.gh_device_info(self._h, name, C.byref(cus), C.byref(mem)))
        return {"name": name.value.decode(), "cus": cus.value, "mem_bytes": mem.value}

    def synchronize(self):
        self._chk(self._lib.gh_synchronize(self._h))

    # -- kernel matrix ------------------------------------------------------------
    def set_obs(self, a, b, c):
        a, b, c = f64(a), f64(b), f64(c)
        if not (a.shape == b.shape == c.shape == (self.N,)):
            raise ValueError("Input arrays xp, yp, and zp must have same length!")
        self._chk(self._lib.gh_set_obs(self._h, ptr(a), ptr(b), ptr(c)))

    def set_cells(self, bounds6, kind, ratio=1.6):
        b = f64(bounds6)
        if b.shape != (self.M, 6):
            raise ValueError("bounds table must be (M, 6)")
        self._chk(self._lib.gh_set_cells(self._h, ptr(b), int(kind), float(ratio)))

    def set_matrix_free(self, on=True, exact=None):
        """Never store G.  exact (tesseroids): True = a far pair's GLQ leaf in the reference's operation
        order (_tesseroid_numba.py:207-222), False = the throughput form (~1e-14 from it); None leaves
        the choice to the environment (GRAVHMC_MF_EXACT, default the throughput form)."""
        self._chk(self._lib.gh_set_matrix_free(self._h, 1 if on else 0))
        if exact is not None:
            self._chk(self._lib.gh_set_matrix_free_exact(self._h, 1 if exact else 0))

    def set_shift_invariant(self, on=True):
        """Regular spherical grids (every cell row a full circle of longitudes, observations on the same
        spacing): keep K[i, (c, k)] = T[c][class_i][(m_i - k) mod n] instead of G (gh_set_shift_invariant);
        build_G raises NotImplementedError with the reason if the geometry lacks the structure."""
        self._chk(self._lib.gh_set_shift_invariant(self._h, 1 if on else 0))
        self._shift_invariant = bool(on)

    def shift_invariant_info(self):
        n, na, nc, tb = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int64(0)
        self._chk(self._lib.gh_shift_invariant_info(self._h, C.byref(n), C.byref(na), C.byref(nc), C.byref(tb)))
        return {"n_lon": n.value, "n_classes": na.value, "n_rows": nc.value, "table_bytes": tb.value}

    def shift_invariant_harmonic(self):
        """The store in the longitude-harmonic domain: on, form ("registers": csrc/lonsymh.hip.h, "streamed":
        csrc/lonsymw.hip.h for grids beyond 126 longitudes / 64 observation classes), frequencies, bytes of T^, workgroups."""
        on, nf, tb, wg = C.c_int(0), C.c_int(0), C.c_int64(0), C.c_int(0)
        self._chk(self._lib.gh_shift_invariant_harmonic(self._h, C.byref(on), C.byref(nf), C.byref(tb), C.byref(wg)))
        return {"on": bool(on.value), "form": {0: None, 1: "registers", 2: "streamed"}[on.value], "n_freq": nf.value,
                "table_bytes": tb.value, "workgroups": wg.value}

    def matrix_free_stats(self):
        """Entries / GLQ leaves evaluated and launches of the fused matrix-free pass since
        profile_enable(True)."""
        e, l, n, ne, nl = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.gh_matrix_free_stats(self._h, C.byref(e), C.byref(l), C.byref(n), C.byref(ne),
                                                 C.byref(nl)))
        return {"entries": e.value, "leaves": l.value, "launches": n.value,
                "near_entries": ne.value, "near_leaves": nl.value}

    def build_G(self):
        self._chk(self._lib.gh_build_G(self._h))

    def kernel_stats(self):
        w, l = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.gh_kernel_stats(self._h, C.byref(w), C.byref(l)))
        return {"warn_cells": w.value, "leaves": l.value}

    def upload_G(self, A):
        A = np.asarray(A, dtype=np.float64)
        if A.shape != (self.N, self.M):
            raise ValueError("kernel must be N x M")
        if A.flags.f_contiguous:
            self._chk(self._lib.gh_upload_G(self._h, ptr(A), self.N, 1))
        else:
            A = np.ascontiguousarray(A)
            self._chk(self._lib.gh_upload_G(self._h, ptr(A), self.M, 0))

    def download_G(self):
        A = np.empty((self.M, self.N))
        self._chk(self._lib.gh_download_G(self._h, ptr(A), self.N))
        return A.T  # N x M, Fortran-ordered view

    def weight(self, weightfactor=0.5):
        wm = np.empty(self.M)
        self._chk(self._lib.gh_weight(self._h, float(weightfactor), ptr(wm)))
        return wm

    # -- potential ----------------------------------------------------------------
    def set_data(self, dobs, grav_fix=None):
        dobs = f64(dobs)
        if dobs.shape != (self.N,):
            raise ValueError("dobs must have N entries")
        gf = f64(grav_fix) if grav_fix is not None else None
        self._chk(self._lib.gh_set_data(self._h, ptr(dobs), ptr(gf)))

    def set_reg(self, regularization, alpha, beta, shape, mwapr):
        if regularization not in _lib.REG_KINDS:
            raise ValueError("Please choose regularization from 'MS','Damping', 'Smoothness', 'TV'.")
        mwapr = f64(mwapr)
        if mwapr.shape != (self.M,):
            raise ValueError("mwapr must have M entries")
        shp = (C.c_int * 3)(*[int(s) for s in shape]) if shape is not None else None
        self._chk(self._lib.gh_set_reg(self._h, _lib.REG_KINDS[regularization], float(alpha),
                                       float(beta), shp, ptr(mwapr)))
        self._chain_valid = False
        self._reg_key = None   # (GravMagModule._use_reg records what it sent after this call)

    def forward(self, mw):
        mw = self._vecM(mw, "mw")
        d = np.empty(self.N)
        self._chk(self._lib.gh_forward(self._h, ptr(mw), ptr(d)))
        return d

    def adjoint(self, r):
        r = self._vecN(r, "r")
        g = np.empty(self.M)
        self._chk(self._lib.gh_adjoint(self._h, ptr(r), ptr(g)))
        return g

    def misfit_and_grad(self, x):
        x = self._vecM(x, "x")
        out3, grad, dpre = np.empty(3), np.empty(self.M), np.empty(self.N)
        self._chk(self._lib.gh_misfit_and_grad(self._h, ptr(x), ptr(out3), ptr(grad), ptr(dpre)))
        return out3[0], grad, dpre, out3[1], out3[2]

    def reg_eval(self, regularization, mw, mwapr, beta=0.01, shape=None, ms_grad_den_mw=False,
                 want_grad=True):
        """(value, grad) of one regulariser alone (alpha = 1)."""
        if regularization not in _lib.REG_KINDS:
            raise ValueError("Please choose regularization from 'MS','Damping', 'Smoothness', 'TV'.")
        mw, mwapr = self._vecM(mw, "mw"), self._vecM(mwapr, "mwapr")
        shp = (C.c_int * 3)(*[int(s) for s in shape]) if shape is not None else None
        val = C.c_double(0)
        grad = np.empty(self.M) if want_grad else None
        self._chk(self._lib.gh_reg_eval(self._h, _lib.REG_KINDS[regularization], float(beta), shp,
                                        1 if ms_grad_den_mw else 0, ptr(mw), ptr(mwapr),
                                        C.byref(val), ptr(grad)))
        return val.value, grad

    # -- wavelet-compressed forward -----------------------------------------------
    def compress_wavelet(self, dims, shape=None, thr=1e-3, levels=2):
        shp = (C.c_int * 3)(*[int(v) for v in shape]) if shape is not None else None
        nnz, ncols = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.gh_compress_wavelet(self._h, int(dims), shp, float(thr), int(levels),
                                                C.byref(nnz), C.byref(ncols)))
        self.wavelet_nnz, self.wavelet_ncols = nnz.value, ncols.value
        return nnz.value, ncols.value

    def download_csr(self):
        from scipy.sparse import csr_matrix
        indptr = np.empty(self.N + 1, dtype=np.int64)
        indices = np.empty(max(self.wavelet_nnz, 1), dtype=np.int32)
        data = np.empty(max(self.wavelet_nnz, 1))
        self._chk(self._lib.gh_download_csr(self._h, indptr.ctypes.data_as(C.POINTER(C.c_int64)),
                                            indices.ctypes.data_as(C.POINTER(C.c_int32)), ptr(data)))
        n = self.wavelet_nnz
        return csr_matrix((data[:n], indices[:n], indptr), shape=(self.N, self.wavelet_ncols))

    def model_coeffs(self, mw):
        mw = self._vecM(mw, "mw")
        out = np.empty(self.wavelet_ncols)
        self._chk(self._lib.gh_model_coeffs(self._h, ptr(mw), ptr(out)))
        return out

    def forward_wavelet(self, mw):
        mw = self._vecM(mw, "mw")
        d = np.empty(self.N)
        self._chk(self._lib.gh_forward_wavelet(self._h, ptr(mw), ptr(d)))
        return d

    # -- chain --------------------------------------------------------------------
    def chain_init(self, x0, low, high):
        x0, low, high = self._vecM(x0, "x0"), self._vecM(low, "low"), self._vecM(high, "high")
        self._chk(self._lib.gh_chain_init(self._h, ptr(x0), ptr(low), ptr(high)))
        self._chain_valid = True

    def chain_trajectory(self, p0, dt, L, u):
        p0 = self._vecM(p0, "p0")
        acc = C.c_int(0)
        out5 = np.empty(5)
        self._chk(self._lib.gh_chain_trajectory(self._h, ptr(p0), float(dt), int(L), float(u),
                                                C.byref(acc), ptr(out5)))
        return bool(acc.value), out5

    def chain_prefetch_momentum(self, p0_next):
        p0_next = self._vecM(p0_next, "p0_next")
        self._chk(self._lib.gh_chain_prefetch_momentum(self._h, ptr(p0_next)))

    def chain_stats(self):
        a, b = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.gh_chain_stats(self._h, C.byref(a), C.byref(b)))
        la, ev = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.gh_chain_resident_stats(self._h, C.byref(la), C.byref(ev)))
        q, tl, to, late = C.c_int(0), C.c_int64(0), C.c_int(0), C.c_int64(0)
        self._chk(self._lib.gh_team_sweep_stats(self._h, C.byref(q), C.byref(tl), C.byref(to), C.byref(late)))
        rb = self.batch_resident_stats()
        return {"spec_hits": a.value, "spec_misses": b.value,
                "resident_launches": la.value, "resident_evaluations": ev.value + rb["lock_steps"],
                "resident_batch_launches": rb["launches"], "resident_batch_lock_steps": rb["lock_steps"],
                "team_members": q.value, "team_launches": tl.value, "team_timeouts": to.value,
                "team_late_parts": late.value}

    def _prepare_batch(self, blk, look, want_x):
        """Arguments of one gh_chain_run call over a block (Ls, p0s[K, M], us) of trajectories,
        marshalled on the calling thread (the 4 MB per momentum of C2 are copied here, not between
        two batches on the GPU).  `look`: the block whose first row is the next trajectory."""
        Ls, p0s, us = blk
        K = len(Ls)
        p0s = self._loc_rows(p0s)
        if p0s.shape != (K, self.M):
            raise ValueError("p0 must have M = %d entries, got shape %r" % (self.M, p0s.shape[1:]))
        return {"K": K, "Ls": np.ascontiguousarray(Ls, dtype=np.int32), "p0s": p0s,
                "us": np.ascontiguousarray(us, dtype=np.float64),
                "look": self._loc_rows(look[1][:1])[0] if look is not None else None,
                "acc": (C.c_int * K)(), "out5": np.empty((K, 5)),
                "xs": np.empty((K, self.M)) if want_x else None, "n_run": C.c_int(0)}

    def _run_batch(self, a, dt, stop_at, record_from, want_x, entered=None):
        """One gh_chain_run call over a prepared block; returns the number of trajectories run.
        `entered` (threading.Event) is set right before the library call (which releases the GIL)."""
        if entered is not None:
            entered.set()
        self._chk(self._lib.gh_chain_run(self._h, a["K"], a["Ls"].ctypes.data_as(C.POINTER(C.c_int)),
                                         ptr(a["p0s"]), ptr(a["us"]), float(dt),
                                         ptr(a["look"]), int(stop_at), int(record_from), a["acc"],
                                         ptr(a["out5"]), ptr(a["xs"]), C.byref(a["n_run"])))
        return a["n_run"].value   # results: a["acc"], a["out5"], a["xs"] (first n_run rows)

    def default_batch(self):
        """Trajectories per gh_chain_run call: enough to hide the per-call cost (Python round trip:
        ~0.5 ms of idle GPU per call at C2; momentum upload; for small problems the launch of the
        resident chain kernel), few enough that the host draw of the next batch still overlaps the
        GPU (C2: 4, C1: 128)."""
        if getattr(self, "_shift_invariant", False):
            # (the shift-invariant store's passes take tens of microseconds: a call's fixed costs -- two extra evaluations
            # of the persistent launch, the staging of the momenta -- want more trajectories to spread over)
            return int(max(4, min(128, (24 << 20) // (8 * max(1, self.M)))))
        return int(max(4, min(128, (6 << 20) // (8 * max(1, self.M)))))

    def _vec_any(self, v):
        """One momentum as a caller hands it over (full length; a sharded engine slices it later)."""
        v = f64(v)
        if v.ndim != 1:
            raise ValueError("p0 must be a vector, got shape %r" % (v.shape,))
        return v

    def _loc_rows(self, p0s):
        """This engine's columns of a block of momenta, contiguous float64 [K, M]."""
        return np.ascontiguousarray(p0s, dtype=np.float64)

    def _full_vec(self, v):
        return v

    def run_chain(self, draws, dt, on_result, stop_at_accepts=0, record_from=0, want_x=False,
                  batch=None, overlap=False):
        """Pipelined trajectories: `draws` yields (L, p0, u) in RNG-stream order.  Batches of
        trajectories run inside one library call (gh_chain_run: momentum of trajectory k+1
        announced before trajectory k, so an accepted proposal's last sweep already takes the
        next first step; small dense problems: one launch of the resident chain kernel per batch)
        while the next batch is drawn on the host.
        `on_result(L, accepted, out5, x)` is called per finished trajectory (x = chain state after
        an accepted trajectory if want_x, else None) and may return False to stop.
        overlap=True hands a finished batch to `on_result` while the next one already runs (the
        device never waits for Python's bookkeeping or file output); a stop requested by
        `on_result` -- as opposed to `stop_at_accepts`, which the library itself honours, or the
        end of `draws` -- then takes effect one batch late."""
        import threading
        if batch is None:
            batch = self.default_batch()
        # The momenta of a batch are drawn / gathered into page-locked blocks of the library (three of them in
        # rotation: the batch that runs, the one being drawn, one spare): the library sends them to the GPU from there
        # -- a copy from ordinary memory goes through a staging buffer first, 23 MB per call at 72 000 cells.
        # (kept with the engine between calls: page-locking tens of MB costs milliseconds)
        pool = self.__dict__.setdefault("_p0_pool", {"i": 0, "blocks": []})

        def p0_block(n, width):
            if n * width * 8 * 3 > (2 << 30) or n < 2:
                return np.empty((n, width))
            if not pool["blocks"] or pool["blocks"][0].shape[1] != width or pool["blocks"][0].shape[0] < n:
                for b in pool["blocks"]:
                    self.pinned_free(b)
                pool["blocks"] = [self.pinned_empty((max(n, batch), width)) for _ in range(3)]
            pool["i"] = (pool["i"] + 1) % 3
            return pool["blocks"][pool["i"]][:n]

        if hasattr(draws, "take_block"):
            # a source that fills blocks itself (inversion.rng.LegacyDraws): rows after the lookahead
            # are drawn straight into the batch's arrays
            def take(n, head=None):
                if head is None:
                    if n < 2:
                        return draws.take_block(n)
                    out = (np.empty(n, dtype=np.int32), p0_block(n, draws.M), np.empty(n))
                    got = len(draws.take_block(n, out=out, at=0)[0])
                    return tuple(o[:got] for o in out)
                out = (np.empty(n, dtype=np.int32), p0_block(n, head[1].shape[1]), np.empty(n))
                out[0][0], out[1][0], out[2][0] = head[0][0], head[1][0], head[2][0]
                got = len(draws.take_block(n - 1, out=out, at=1)[0]) if n > 1 else 0
                return tuple(o[:1 + got] for o in out)
        else:
            it = iter(draws)

            def take(n, head=None):
                rows = []
                for _ in range(n - (0 if head is None else 1)):
                    d = next(it, None)
                    if d is None:
                        break
                    rows.append(d)
                Ls = [int(head[0][0])] if head is not None else []
                ps = [head[1][0]] if head is not None else []
                us = [float(head[2][0])] if head is not None else []
                Ls += [int(d[0]) for d in rows]
                ps += [self._vec_any(d[1]) for d in rows]
                us += [float(d[2]) for d in rows]
                if not Ls:
                    return (np.empty(0, dtype=np.int32), np.empty((0, 0)), np.empty(0))
                blk = p0_block(len(ps), ps[0].shape[0])
                for i, pv in enumerate(ps):
                    blk[i] = pv
                return (np.asarray(Ls, dtype=np.int32), blk, np.asarray(us, dtype=np.float64))

        # one worker thread for the whole chain takes the library calls (a thread per call costs
        # ~0.1 ms of idle device between two batches of a small problem)
        import queue
        jobs = queue.Queue()

        def worker():
            while True:
                item = jobs.get()
                if item is None:
                    return
                prepared, res, entered, done = item
                try:
                    res["n"] = self._run_batch(prepared, dt, stop_at_accepts, record_from, want_x, entered)
                except BaseException as e:  # re-raised in the caller's thread
                    res["e"] = e
                finally:
                    entered.set()
                    done.set()

        th = threading.Thread(target=worker, daemon=True)
        th.start()

        def start(prepared):
            res = {"a": prepared}
            entered, done = threading.Event(), threading.Event()
            jobs.put((prepared, res, entered, done))
            # the GPU has its work before this thread goes back to drawing (np.random's generator
            # holds the GIL for the ~10 ms a C2 momentum takes)
            entered.wait()
            return done, res

        def finish(job):
            job[0].wait()
            if "e" in job[1]:
                raise job[1]["e"]
            return job[1]["n"], job[1]["a"]

        def some(blk):
            return blk is not None and len(blk[0]) > 0

        try:
            # a momentum of >= 1 MB takes milliseconds to draw: the first call then carries ONE
            # trajectory, so the device starts after two draws instead of batch + 1
            cur = take(1 if self.M * 8 >= (1 << 20) else batch)
            look = take(1) if some(cur) else None
            look = look if some(look) else None
            job = start(self._prepare_batch(cur, look, want_x)) if some(cur) else None
            while some(cur):
                # the lookahead trajectory opens the next batch; the rest is drawn and marshalled
                # while the GPU runs
                nxt = take(batch, head=look) if look is not None else None
                nlook = take(1) if some(nxt) else None
                nlook = nlook if some(nlook) else None
                nprepared = self._prepare_batch(nxt, nlook, want_x) if some(nxt) else None
                n_run, a = finish(job)
                short = n_run < len(cur[0])               # the library stopped at stop_at_accepts
                job = start(nprepared) if (overlap and some(nxt) and not short) else None
                stop = False
                acc, out5, xs = a["acc"], a["out5"], a["xs"]
                for k in range(n_run):
                    x = xs[k].copy() if (want_x and acc[k]) else None
                    if on_result(int(cur[0][k]), bool(acc[k]), out5[k],
                                 self._full_vec(x) if x is not None else None) is False:
                        stop = True
                        break
                if stop or short or not some(nxt):
                    if job is not None:
                        finish(job)
                    break
                if job is None:
                    job = start(nprepared)
                cur, look = nxt, nlook
        finally:
            jobs.put(None)
            th.join()

    def chain_get_x(self):
        x = np.empty(self.M)
        self._chk(self._lib.gh_chain_get_x(self._h, ptr(x)))
        return x

    def chain_get_dsyn(self):
        d = np.empty(self.N)
        self._chk(self._lib.gh_chain_get_dsyn(self._h, ptr(d)))
        return d

    # -- several chains per GPU (MFMA) ----------------------------------------------
    def batch_init(self, x0s, low, high):
        x0s = np.ascontiguousarray(np.atleast_2d(x0s), dtype=np.float64)
        if x0s.ndim != 2 or x0s.shape[1] != self.M:
            raise ValueError("x0s must be (C, M)")
        self._batch_C = x0s.shape[0]
        low, high = self._vecM(low, "low"), self._vecM(high, "high")
        self._chk(self._lib.gh_batch_init(self._h, self._batch_C, ptr(x0s), ptr(low), ptr(high)))

    def batch_trajectory(self, p0s, dt, Ls, us):
        Cn = self._batch_C
        p0s = np.ascontiguousarray(np.atleast_2d(p0s), dtype=np.float64)
        if p0s.shape != (Cn, self.M):
            raise ValueError("p0s must be (C, M)")
        Ls_ = (C.c_int * Cn)(*[int(v) for v in Ls])
        us_ = np.ascontiguousarray(us, dtype=np.float64)
        acc = (C.c_int * Cn)()
        out5 = np.empty((Cn, 5))
        self._chk(self._lib.gh_batch_trajectory(self._h, ptr(p0s), float(dt), Ls_, ptr(us_), acc,
                                                ptr(out5)))
        return [bool(a) for a in acc], out5

    def batch_run(self, p0s, dt, Ls, us, want_x=False, carry=False, entered=None):
        """Up to T further trajectories of every chain, the chains desynchronised (gh_batch_run).
        p0s: (C, T, M) array, or C lists of T momentum vectors (not copied); Ls (C, T), us (C, T):
        the trajectories each chain has not started yet.  Returns accepted (C, S) bool, out5
        (C, S, 5), xs (C, S, M) or None -- per chain in order of completion, S = T slots, T + 1
        with carry -- and, with carry=True, (n_started, n_done): the call then ends as soon as a chain has nothing left to
        start, the others keep their trajectory in flight for the next call (T = 0, i.e. C empty
        lists, drains them).  `entered` (threading.Event) is set right before the library call, which releases the
        interpreter lock: a caller that runs this on a worker thread waits for it before it starts drawing the next
        offers on a dozen threads, so the GPU has its work first."""
        Cn = len(p0s)
        T = len(p0s[0]) if Cn else 0
        if any(len(r) != T for r in p0s):
            raise ValueError("p0s must hold the same number of M-vectors for every chain")
        if T and isinstance(p0s[0][0], (int, np.integer)):
            # row ADDRESSES (LegacyDraws.take_ring: M float64 values each, alive for the call)
            addr = np.ascontiguousarray(p0s, dtype=np.uint64).reshape(Cn * T)
            ptrs = addr.ctypes.data_as(C.POINTER(_lib._dp))
        else:
            rows = [[f64(p) for p in chain] for chain in p0s]
            if any(p.shape != (self.M,) for r in rows for p in r):
                raise ValueError("p0s must hold the same number of M-vectors for every chain")
            ptrs = (_lib._dp * max(1, Cn * T))(*[ptr(p) for r in rows for p in r])
        To = T + 1 if carry else T            # result slots per chain
        Ls = np.ascontiguousarray(Ls, dtype=np.int32).reshape(Cn, T)
        us = np.ascontiguousarray(us, dtype=np.float64).reshape(Cn, T)
        acc = np.zeros((Cn, To), dtype=np.int32)
        out5 = np.zeros((Cn, To, 5))
        xs = np.empty((Cn, To, self.M)) if want_x else None
        ns = np.zeros(Cn, dtype=np.int32)
        nd = np.zeros(Cn, dtype=np.int32)
        ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
        if entered is not None:
            entered.set()
        self._chk(self._lib.gh_batch_run(self._h, int(T), ip(Ls) if T else None, ptrs if T else None,
                                         ptr(us) if T else None, float(dt), ip(acc), ptr(out5), ptr(xs),
                                         ip(ns) if carry else None, ip(nd) if carry else None))
        if carry:
            return acc.astype(bool), out5, xs, ns, nd
        return acc.astype(bool), out5, xs

    def shift_invariant_resident_stats(self):
        """The harmonic pass of the shift-invariant store as one persistent launch per run_chain call
        (csrc/lonres.hip.h): grid, launches, evaluations, trajectories, time-outs."""
        wg, t = C.c_int(0), C.c_int(0)
        l, e, tr = C.c_int64(0), C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.gh_shift_invariant_resident_stats(self._h, C.byref(wg), C.byref(l), C.byref(e), C.byref(tr),
                                                              C.byref(t)))
        return {"workgroups": wg.value, "launches": l.value, "evaluations": e.value, "trajectories": tr.value,
                "timeouts": t.value}

    def pinned_empty(self, shape):
        """A float64 array in page-locked host memory of the library (gh_pinned_alloc): momentum rows drawn into
        it go to the device without a gather on the host.  It lives until pinned_free(array) or close()."""
        n = int(np.prod(shape))
        p = C.c_void_p()
        self._chk(self._lib.gh_pinned_alloc(self._h, max(1, n) * 8, C.byref(p)))
        buf = (C.c_double * max(1, n)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=np.float64, count=n).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def pinned_free(self, arr):
        p = getattr(self, "_pinned", {}).pop(arr.ctypes.data, None)
        if p is None:
            raise ValueError("not an array of pinned_empty")
        self._chk(self._lib.gh_pinned_free(self._h, C.c_void_p(p)))

    def batch_staging_stats(self):
        """Momentum rows of the lock-step form sent straight from pinned memory / gathered on the host first."""
        d, s = C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.gh_batch_staging_stats(self._h, C.byref(d), C.byref(s)))
        return {"rows_direct": d.value, "rows_staged": s.value}

    def batch_fused_stats(self):
        """Team form of the matrix-free batch (one evaluation per entry and step): grid, launches, time-outs."""
        m, r, l, t = C.c_int(0), C.c_int(0), C.c_int64(0), C.c_int(0)
        self._chk(self._lib.gh_batch_fused_stats(self._h, C.byref(m), C.byref(r), C.byref(l), C.byref(t)))
        return {"members": m.value, "ranges": r.value, "launches": l.value, "timeouts": t.value}

    def batch_resident_stats(self):
        """Chains in lock-step inside the resident batch kernel (csrc/resbatch.hip.h): launches, lock-steps,
        evaluations of all chains, lock-steps lost to rejected speculative first steps, time-outs."""
        l, s, cs, lo, t = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
        self._chk(self._lib.gh_batch_resident_stats(self._h, C.byref(l), C.byref(s), C.byref(cs), C.byref(lo),
                                                    C.byref(t)))
        return {"launches": l.value, "lock_steps": s.value, "chain_steps": cs.value, "lost_steps": lo.value,
                "timeouts": t.value}

    def matrix_free_team_stats(self):
        """Team form of the single-chain matrix-free pass: grid, launches, time-outs."""
        m, r, l, t = C.c_int(0), C.c_int(0), C.c_int64(0), C.c_int(0)
        self._chk(self._lib.gh_matrix_free_team_stats(self._h, C.byref(m), C.byref(r), C.byref(l), C.byref(t)))
        return {"members": m.value, "ranges": r.value, "launches": l.value, "timeouts": t.value}

    def batch_get_x(self, chain):
        x = np.empty(self.M)
        self._chk(self._lib.gh_batch_get_x(self._h, int(chain), ptr(x)))
        return x

    # -- posterior window ---------------------------------------------------------
    def posterior_window(self, K=100):
        self._chk(self._lib.gh_posterior_window(self._h, int(K)))
        self._has_window = True

    def posterior_add(self):
        self._chk(self._lib.gh_posterior_add(self._h))

    def posterior_read(self, want_arrays=True):
        n, tot = C.c_int64(0), C.c_int64(0)
        mean = np.empty(self.M) if want_arrays else None
        sd = np.empty(self.M) if want_arrays else None
        self._chk(self._lib.gh_posterior_read(self._h, C.byref(n), C.byref(tot), ptr(mean), ptr(sd)))
        return {"n": n.value, "total": tot.value, "mean": mean, "std": sd}

    def leapfrog(self, x, p0, dt, L, low, high, u, want_dsyn=True):
        x = self._vecM(x, "x").copy()
        p0, low, high = self._vecM(p0, "p0"), self._vecM(low, "low"), self._vecM(high, "high")
        acc = C.c_int(0)
        out5 = np.empty(5)
        dsyn = np.empty(self.N) if want_dsyn else None
        self._chk(self._lib.gh_leapfrog(self._h, ptr(x), ptr(p0), float(dt), int(L), ptr(low),
                                        ptr(high), float(u), C.byref(acc), ptr(out5), ptr(dsyn)))
        return x, bool(acc.value), out5, dsyn

    # -- measurement --------------------------------------------------------------
    def profile_enable(self, on=True):
        self._chk(self._lib.gh_profile_enable(self._h, 1 if on else 0))

    def stream_read_gbps(self, nt=True, reps=3):
        """Attainable streaming-read rate (GB/s) of this device over the stored kernel matrix."""
        ms = C.c_double(0)
        self._chk(self._lib.gh_measure_stream_read(self._h, 1 if nt else 0, int(reps), C.byref(ms)))
        return self.N * self.M * 8 / (ms.value * 1e-3) / 1e9

    def profile_read(self):
        ms, n, b = C.c_double(0), C.c_int64(0), C.c_int64(0)
        self._chk(self._lib.gh_profile_read(self._h, C.byref(ms), C.byref(n), C.byref(b)))
        return {"sweep_ms": ms.value, "sweeps": n.value, "bytes_per_sweep": b.value}
