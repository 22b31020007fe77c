"""The reference's random draws per trajectory, taken from libgravhmc instead of np.random.

HamitonianMC draws, per trajectory and in this order, ``L = np.random.randint(Lmin, Lmax + 1)``
(reference hmc.py:297), ``p0 = np.random.randn(M) * Sigma`` (hmc.py:95) and ``u = np.random.rand()``
(hmc.py:164) from NumPy's global legacy generator.  At the small configurations (C1, C3: 6000
cells) those 6000 normals cost as much host time as the GPU needs for the whole trajectory.
``gh_rng_draw_trajectories`` restates the legacy stream bit for bit and scales the normals on
several threads; :class:`LegacyDraws` adopts the global generator's state, hands out blocks of
trajectories in the arrays ``gh_chain_run`` takes, and gives the state back on ``release()``, so
code before and after the chain sees the stream the reference would have left (up to the
trajectories the pipeline drew ahead, as before).
"""
import ctypes as C

import numpy as np

from .. import _lib


def draw_workers(n_chains):
    """Threads for the draws of n_chains generators side by side: a draw is mostly SEQUENTIAL (the twists of MT19937
    and the polar method's rejections: 3.5 ns per normal on one core of a GPU box's host, the logarithms included), so the
    chains' draws go on as many cores as the process may use less three (the thread that feeds the GPU, the one that
    reports, the interpreter) -- one generator per core, no helpers."""
    return int(max(1, min(int(n_chains), _lib.load().gh_host_cores() - 3)))


class LegacyDraws(object):
    """Iterator of (L, p0, u) in the reference's stream order, with a block interface for
    Engine.run_chain (``take_block``)."""

    def __init__(self, M, Lrange, Sigma, fixed_L=None, limit=None, seed=None, helpers=None):
        """seed=None: adopt (and on release() hand back) the state of NumPy's GLOBAL legacy generator, as
        HamitonianMC.sample uses it.  seed=int: an independent stream, bit for bit the one
        ``np.random.RandomState(seed)`` produces (a chain of HMCSampleBatch: the reference's rank seeds its
        global generator with seed + rank, hmc.py:369); release() then only frees it."""
        self._lib = _lib.load()
        self.M = int(M)
        self.Lmin, self.Lmax = int(Lrange[0]), int(Lrange[1])
        sig = np.asarray(Sigma, dtype=np.float64)
        self._sigma = float(sig) if sig.ndim == 0 else 1.0
        self._sigma_vec = None if sig.ndim == 0 else np.ascontiguousarray(np.broadcast_to(sig, (self.M,)))
        self._plan = None if fixed_L is None else list(fixed_L)   # trajectory lengths given, no randint
        self._left = limit if self._plan is None else len(self._plan)
        self._h = C.c_void_p()
        self._own = seed is not None
        if self._lib.gh_rng_create(C.byref(self._h), int(seed) if self._own else 0) != 0:
            raise RuntimeError("gh_rng_create failed")
        if not self._own:
            self.adopt()
            # (ONE chain on the global stream: its draws are what the GPU waits for -- at 72 000 cells a trajectory's
            # normals cost 250 us on one core against 250 us of GPU time -- so the scale pass runs on a second thread
            # behind the sequential generation; the seeded generators of a batch of chains draw side by side, one per core)
            self._lib.gh_rng_set_threads(self._h, -1)
        if helpers is not None:
            self._lib.gh_rng_set_threads(self._h, int(helpers))
        self._row = None

    # -- exchange with np.random --------------------------------------------------------
    def adopt(self):
        kind, key, pos, has_gauss, cached = np.random.get_state()
        if kind != "MT19937":
            raise RuntimeError("np.random is not the legacy MT19937 generator")
        key = np.ascontiguousarray(key, dtype=np.uint32)
        rc = self._lib.gh_rng_set_state(self._h, key.ctypes.data_as(C.POINTER(C.c_uint32)), int(pos),
                                        int(has_gauss), float(cached))
        if rc != 0:
            raise RuntimeError("gh_rng_set_state failed")

    def release(self):
        """Hand the stream back to np.random and free the native generator (and the ring of use_ring)."""
        if getattr(self, "_ring", None) is not None:
            ring, self._ring = self._ring, None
            try:
                self._ring_engine.pinned_free(ring)
            except Exception:
                pass                      # (the engine was closed first: its blocks went with it)
        if not self._h:
            return
        if self._own:
            self._lib.gh_rng_destroy(self._h)
            self._h = C.c_void_p()
            return
        key = np.zeros(624, dtype=np.uint32)
        pos, has_gauss, cached = C.c_int(), C.c_int(), C.c_double()
        self._lib.gh_rng_get_state(self._h, key.ctypes.data_as(C.POINTER(C.c_uint32)), C.byref(pos),
                                   C.byref(has_gauss), C.byref(cached))
        np.random.set_state(("MT19937", key, pos.value, has_gauss.value, cached.value))
        self._lib.gh_rng_destroy(self._h)
        self._h = C.c_void_p()

    def __del__(self):
        try:
            if self._h:
                self._lib.gh_rng_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    # -- draws --------------------------------------------------------------------------
    def take_block(self, n, out=None, at=0):
        """Up to n trajectories: (Ls int32[k], p0s float64[k, M], us float64[k]); k < n at the end of
        a bounded stream.  With `out` = (Ls, p0s, us) arrays of a larger block they are written
        from row `at` on and views of those rows are returned."""
        if self._left is not None:
            n = min(n, self._left)
        if out is None:
            Ls, p0s, us = np.empty(n, dtype=np.int32), np.empty((n, self.M)), np.empty(n)
        else:
            Ls, p0s, us = out[0][at:at + n], out[1][at:at + n], out[2][at:at + n]
        if n <= 0:
            return Ls[:0], p0s[:0], us[:0]
        ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
        if self._plan is None:
            rc = self._lib.gh_rng_draw_trajectories(self._h, n, self.Lmin, self.Lmax, self.M, self._sigma,
                                                    Ls.ctypes.data_as(ip), p0s.ctypes.data_as(dp),
                                                    us.ctypes.data_as(dp))
        else:
            # lengths fixed by the caller: Lmin == Lmax draws nothing from the stream
            rc = self._lib.gh_rng_draw_trajectories(self._h, n, 0, 0, self.M, self._sigma,
                                                    Ls.ctypes.data_as(ip), p0s.ctypes.data_as(dp),
                                                    us.ctypes.data_as(dp))
            done = len(self._plan) - self._left
            Ls[:] = self._plan[done:done + n]
        if rc != 0:
            raise RuntimeError("gh_rng_draw_trajectories failed (%d)" % rc)
        if self._sigma_vec is not None:
            p0s *= self._sigma_vec
        if self._left is not None:
            self._left -= n
        return Ls, p0s, us

    # -- a ring of page-locked rows -------------------------------------------------------
    def use_ring(self, engine, rows):
        """Draw into a ring of `rows` momentum rows in page-locked memory of the engine's library
        (gh_pinned_alloc): Engine.batch_run sends such rows to the device from where they lie, a chain's adjacent
        rows in one copy.  take_ring(n) writes the next n rows; the caller keeps at most `rows` of them in use
        (drawn and not yet started by a call that has returned), oldest first."""
        self._ring_engine = engine
        self._ring = engine.pinned_empty((int(rows), self.M))
        self._ring_Ls = np.empty(int(rows), dtype=np.int32)
        self._ring_us = np.empty(int(rows))
        self._ring_head = 0
        self._ring_base = self._ring.ctypes.data
        return self

    def take_ring(self, n):
        """Up to n further trajectories into the ring: a list of (L, row address, u, row index).  The address is
        what Engine.batch_run takes in place of an array; ring_row(index) is the row as an array."""
        rows = self._ring.shape[0]
        if n > rows:
            raise ValueError("take_ring: more rows than the ring holds")
        out, stride = [], self.M * 8
        while n > 0:
            h = self._ring_head
            k = min(n, rows - h)
            Ls, _, us = self.take_block(k, out=(self._ring_Ls, self._ring, self._ring_us), at=h)
            got = len(Ls)
            base = self._ring_base + h * stride
            out.extend((int(Ls[i]), base + i * stride, float(us[i]), h + i) for i in range(got))
            self._ring_head = (h + got) % rows
            if got < k:
                break
            n -= k
        return out

    def ring_row(self, index):
        return self._ring[index]

    def __iter__(self):
        return self

    def __next__(self):
        Ls, p0s, us = self.take_block(1)
        if len(Ls) == 0:
            raise StopIteration
        return int(Ls[0]), p0s[0], float(us[0])
