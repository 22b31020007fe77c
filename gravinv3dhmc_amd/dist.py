"""One-process-per-GPU launch helpers (the reference's `mpiexec -n K python main_*.py`).

The reference's only use of MPI is `Get_rank()/Get_size()` (example/*/main_*.py:20-22):
every rank is an independent chain with seed + rank and its own output folder
(hmc.py:367-369) -- there is no exchange step on the data path, so none is added here.
`torch.distributed` (gloo) is used purely as a control plane: barrier, max of timings and
gathering per-chain summaries on rank 0.  Launch with
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 your_script.py
"""
import os


class Ranks(object):
    """Rank bookkeeping of an N-chain job; degenerates to a single chain without a launcher."""

    def __init__(self, init=True):
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self._dist = None
        if init and self.world > 1:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if not dist.is_initialized():
                dist.init_process_group(backend="gloo", rank=self.rank, world_size=self.world)
            self._dist = dist

    @property
    def device(self):
        """GPU ordinal of this rank (one process per GPU)."""
        return self.local_rank

    def chain_seed(self, seed):
        return seed + self.rank          # hmc.py:369

    def chain_folder(self, save_folder):
        return save_folder + str(self.rank)   # hmc.py:368

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, value):
        """Max of a Python float over ranks (timing of the slowest chain)."""
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t[0])

    def sum(self, value):
        if self._dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t[0])

    def broadcast_bytes(self, raw, src=0):
        """Byte string of rank `src` on every rank (e.g. the RCCL unique id)."""
        if self._dist is None:
            return raw
        box = [raw if self.rank == src else None]
        self._dist.broadcast_object_list(box, src=src)
        return box[0]

    def allreduce_array(self, arr):
        """Element-wise sum over ranks of a float64 numpy array (returns a new array)."""
        import numpy as np
        if self._dist is None:
            return np.array(arr, dtype=np.float64)
        import torch
        t = torch.from_numpy(np.array(arr, dtype=np.float64))
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return t.numpy()

    def gather(self, obj):
        """List of every rank's picklable `obj` on rank 0 (None elsewhere): e.g. posterior
        mean/std of each chain for the multi-chain statistics of plot_real_multichain.py:64-77."""
        if self._dist is None:
            return [obj]
        out = [None] * self.world if self.rank == 0 else None
        self._dist.gather_object(obj, out, dst=0)
        return out

    def close(self):
        if self._dist is not None and self._dist.is_initialized():
            self._dist.destroy_process_group()
            self._dist = None


def run_chains(model_factory, sample_kwargs, ranks=None):
    """Build this rank's model on its GPU and run its chain (HMCSample) -- the body of the
    reference drivers' main() (example/uniformgrid/main_uniform.py:17-88)."""
    from .inversion.hmc import HMCSample
    ranks = ranks or Ranks()
    model = model_factory(ranks.device)
    kw = dict(sample_kwargs)
    kw["myrank"] = ranks.rank
    chain = HMCSample(model, **kw)
    ranks.barrier()
    return chain


# ---------------------------------------------------------------------------------------------
# One chain sharded over several GPUs: column blocks of G (BASELINE config C5)
# ---------------------------------------------------------------------------------------------

def column_partition(M, world, align=1):
    """Contiguous, near-equal split of M cells over `world` ranks: list of (m0, m1).  align > 1:
    every block is a multiple of `align` cells (whole z-planes of the mesh: what the
    Smoothness/TV stencils need when the cells are sharded)."""
    M, world, align = int(M), int(world), max(1, int(align))
    if M % align:
        raise ValueError("%d cells are not a multiple of the alignment %d" % (M, align))
    base, rem = divmod(M // align, world)
    out, m0 = [], 0
    for g in range(world):
        m1 = m0 + (base + (1 if g < rem else 0)) * align
        out.append((m0, m1))
        m0 = m1
    return out


def allgather_slices(ranks, local, parts):
    """Concatenate every rank's contiguous slice (lengths from `parts`) into the full vector."""
    import numpy as np
    if ranks._dist is None:
        return np.asarray(local, dtype=np.float64)
    import torch
    n = max(m1 - m0 for m0, m1 in parts)
    buf = torch.zeros(n, dtype=torch.float64)
    buf[: len(local)] = torch.from_numpy(np.ascontiguousarray(local, dtype=np.float64))
    outs = [torch.zeros(n, dtype=torch.float64) for _ in range(ranks.world)]
    ranks._dist.all_gather(outs, buf)
    return np.concatenate([o.numpy()[: m1 - m0] for o, (m0, m1) in zip(outs, parts)])


def _engine_base():
    from .engine import Engine
    return Engine


def make_sharded_engine(N, M, ranks, device=None, backend="rccl", align=1, axis="cells"):
    """axis="cells" (default): column blocks, below.  axis="rows": row blocks (make_row_sharded_engine).

    Engine-compatible object whose methods take and return FULL model vectors while the
    device holds only this rank's cells.  backend: "rccl" (all-reduce on the GPU stream over
    xGMI) or "gloo" (host-staged through torch.distributed; several ranks per GPU, tests).
    align: cells per z-plane (ny*nx) to shard in whole planes, which the Smoothness/TV
    regularisers need (their stencil crosses the shard boundaries: one boundary plane per
    neighbour travels with the forward partial's all-reduce)."""
    if axis == "rows":
        return make_row_sharded_engine(N, M, ranks, device=device, backend=backend)
    import ctypes as C
    import numpy as np
    from . import _lib
    Engine = _engine_base()

    class ShardedEngine(Engine):
        def __init__(self):
            self.ranks = ranks
            self.parts = column_partition(M, ranks.world, align)
            if min(m1 - m0 for m0, m1 in self.parts) <= 0:
                raise ValueError("more ranks than z-planes: a rank would hold no cells")
            self.m0, self.m1 = self.parts[ranks.rank]
            self.M_global = int(M)
            Engine.__init__(self, N, self.m1 - self.m0, ranks.device if device is None else device)
            self.M_local = self.M
            lib = self._lib
            if backend == "rccl":
                # (RCCL prints a version banner to stdout when it initialises: keep stdout for the
                # caller's own output -- bench.py's single JSON line -- and send the banner to stderr)
                import sys
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)
                try:
                    idbuf = C.create_string_buffer(128)
                    rc = lib.gh_shard_unique_id(idbuf) if ranks.rank == 0 else 0
                    _lib.check(rc, None)
                    raw = ranks.broadcast_bytes(idbuf.raw)
                    idbuf = C.create_string_buffer(raw, 128)
                    rc = lib.gh_shard_init(self._h, idbuf, ranks.rank, ranks.world, self.M_global, self.m0)
                finally:
                    os.dup2(saved, 1)
                    os.close(saved)
                self._chk(rc)
            else:
                def _cb(_user, ptr_, count):
                    try:
                        arr = np.ctypeslib.as_array(ptr_, shape=(count,))
                        arr[:] = ranks.allreduce_array(arr)
                        return 0
                    except Exception:  # reported by the library as GH_ERR_COMM
                        return 1

                self._cb = _lib.ALLREDUCE_FN(_cb)
                self._chk(lib.gh_shard_init_callback(self._h, C.cast(self._cb, C.c_void_p), None,
                                                     ranks.rank, ranks.world, self.M_global, self.m0))

        # -- marshalling helpers ---------------------------------------------------------
        def _loc(self, v):
            v = np.asarray(v, dtype=np.float64)
            if v.shape[0] == self.M_local and self.M_local != self.M_global:
                return v
            if v.shape[0] != self.M_global:
                raise ValueError("expected a model vector of %d entries" % self.M_global)
            return np.ascontiguousarray(v[self.m0:self.m1])

        def _full(self, local):
            return allgather_slices(self.ranks, local, self.parts)

        # -- overrides: full vectors in, full vectors out -----------------------------------
        def set_cells(self, bounds6, kind, ratio=1.6):
            b = np.asarray(bounds6, dtype=np.float64)
            if b.shape[0] == self.M_global:
                b = b[self.m0:self.m1]
            Engine.set_cells(self, b, kind, ratio)

        def weight(self, weightfactor=0.5):
            return self._full(Engine.weight(self, weightfactor))

        def set_reg(self, regularization, alpha, beta, shape, mwapr):
            # Smoothness / TV: the GLOBAL mesh shape; the library checks the plane alignment
            stencil = regularization in ("Smoothness", "TV")
            Engine.set_reg(self, regularization, alpha, beta, shape if stencil else None, self._loc(mwapr))

        def forward(self, mw):
            return Engine.forward(self, self._loc(mw))

        def adjoint(self, r):
            return self._full(Engine.adjoint(self, r))

        def misfit_and_grad(self, x):
            m, g, d, dv, mv = Engine.misfit_and_grad(self, self._loc(x))
            return m, self._full(g), d, dv, mv

        def chain_init(self, x0, low, high):
            Engine.chain_init(self, self._loc(x0), self._loc(low), self._loc(high))

        def chain_trajectory(self, p0, dt, L, u):
            return Engine.chain_trajectory(self, self._loc(p0), dt, L, u)

        def chain_prefetch_momentum(self, p0_next):
            Engine.chain_prefetch_momentum(self, self._loc(p0_next))

        def chain_get_x(self):
            return self._full(Engine.chain_get_x(self))

        def _full_vec(self, v):
            return self._full(v)

        def _loc_rows(self, p0s):
            p0s = np.asarray(p0s, dtype=np.float64)
            if p0s.shape[1] == self.M_local and self.M_local != self.M_global:
                return np.ascontiguousarray(p0s)
            if p0s.shape[1] != self.M_global:
                raise ValueError("expected momenta of %d entries" % self.M_global)
            return np.ascontiguousarray(p0s[:, self.m0:self.m1])

        def run_chain(self, draws, dt, on_result, **kw):
            # Host-staged (gloo) all-reduce: the library's per-step callback and the all_gather of
            # _full_vec use ONE process group; issued from two threads their order would differ
            # across ranks (mismatched collectives).  No batch may run while results are gathered.
            if backend != "rccl":
                kw["overlap"] = False
            return Engine.run_chain(self, draws, dt, on_result, **kw)

        def posterior_read(self, want_arrays=True):
            out = Engine.posterior_read(self, want_arrays)
            if want_arrays:
                out["mean"], out["std"] = self._full(out["mean"]), self._full(out["std"])
            return out

        def download_G(self):
            raise NotImplementedError("the sharded kernel is never gathered on one host")

    return ShardedEngine()


def make_row_sharded_engine(N, M, ranks, device=None, backend="rccl"):
    """Row blocks (BASELINE configs[4] as worded: "G row-block sharded ... RCCL reduce ... for the misfit sum";
    SURVEY 8e.2): the device holds this rank's OBSERVATIONS (rows of G) and all cells.  The object takes and
    returns FULL vectors: observation vectors are sliced on the way in and gathered on the way out, model
    vectors are replicated.  Per evaluation two scalar all-reduces (mean of the predicted data, |r|^2) and an
    all-reduce of the M-vector gradient; two reads of the local shard per leapfrog step (include/gravhmc.h,
    gh_shard_init_rows)."""
    import ctypes as C
    import numpy as np
    from . import _lib
    Engine = _engine_base()

    class RowShardedEngine(Engine):
        def __init__(self):
            self.ranks = ranks
            self.parts = column_partition(N, ranks.world)      # (the same near-equal split, of the rows)
            self.n0, self.n1 = self.parts[ranks.rank]
            self.N_global, self.M_global = int(N), int(M)
            Engine.__init__(self, self.n1 - self.n0, M, ranks.device if device is None else device)
            self.M_local = self.M
            self._backend = backend
            self._sharded = False
            self._shard()

        def _shard(self):
            """gh_shard_init_rows (collective)."""
            if self._sharded:
                return
            lib = self._lib
            if self._backend == "rccl":
                import sys
                sys.stdout.flush()
                saved = os.dup(1)
                os.dup2(2, 1)      # (RCCL's banner goes to stderr: stdout carries the caller's JSON line)
                try:
                    idbuf = C.create_string_buffer(128)
                    rc = lib.gh_shard_unique_id(idbuf) if ranks.rank == 0 else 0
                    _lib.check(rc, None)
                    raw = ranks.broadcast_bytes(idbuf.raw)
                    idbuf = C.create_string_buffer(raw, 128)
                    rc = lib.gh_shard_init_rows(self._h, idbuf, ranks.rank, ranks.world, self.N_global, self.n0)
                finally:
                    os.dup2(saved, 1)
                    os.close(saved)
                self._chk(rc)
            else:
                def _cb(_user, ptr_, count):
                    try:
                        arr = np.ctypeslib.as_array(ptr_, shape=(count,))
                        arr[:] = ranks.allreduce_array(arr)
                        return 0
                    except Exception:
                        return 1

                self._cb = _lib.ALLREDUCE_FN(_cb)
                self._chk(lib.gh_shard_init_rows_callback(self._h, C.cast(self._cb, C.c_void_p), None, ranks.rank,
                                                          ranks.world, self.N_global, self.n0))
            self._sharded = True

        def _rows(self, v):
            v = np.asarray(v, dtype=np.float64)
            if v.shape[0] == self.N_global:
                return np.ascontiguousarray(v[self.n0:self.n1])
            if v.shape[0] != self.n1 - self.n0:
                raise ValueError("expected an observation vector of %d entries" % self.N_global)
            return v

        def _gather_rows(self, local):
            return allgather_slices(self.ranks, local, self.parts)

        def set_obs(self, a, b, c):
            Engine.set_obs(self, self._rows(a), self._rows(b), self._rows(c))

        def upload_G(self, A):
            A = np.asarray(A)
            Engine.upload_G(self, A[self.n0:self.n1] if A.shape[0] == self.N_global else A)

        def set_data(self, dobs, grav_fix=None):
            Engine.set_data(self, self._rows(dobs), None if grav_fix is None else self._rows(grav_fix))

        def forward(self, mw):
            return self._gather_rows(Engine.forward(self, mw))

        def adjoint(self, r):
            return Engine.adjoint(self, self._rows(r))

        def misfit_and_grad(self, x):
            m, g, d, dv, mv = Engine.misfit_and_grad(self, x)
            return m, g, self._gather_rows(d), dv, mv

        def chain_get_dsyn(self):
            return self._gather_rows(Engine.chain_get_dsyn(self))

        def run_chain(self, draws, dt, on_result, **kw):
            if self._backend != "rccl":
                kw["overlap"] = False      # (one process group: no collective from a second thread)
            return Engine.run_chain(self, draws, dt, on_result, **kw)

        def download_G(self):
            raise NotImplementedError("the sharded kernel is never gathered on one host")

    return RowShardedEngine()
