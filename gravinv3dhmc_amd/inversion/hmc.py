"""Hamiltonian Monte Carlo sampler whose trajectories run on the GPU.

Host-side mirror of the reference's `inversion.hmc` (inversion/hmc.py:29-403): class
`HamitonianMC` and function `HMCSample` with the reference's names, arguments, console
lines and sample-file formats.  The random stream stays NumPy's legacy global generator in
the reference's draw order (randint for L -> randn(M) for the momentum -> rand() for the
Metropolis test, hmc.py:297,95,164), so accepted-sample sequences can be compared with the
reference run for run.

The leapfrog loop itself (hmc.py:85-177) is one call into libgravhmc per trajectory
(`gh_chain_trajectory`): the chain state lives in HBM, each leapfrog step is one fused sweep
of the kernel matrix.
"""
import os
import sys

import numpy as np
from scipy.sparse import coo_matrix

from ..utils import write_rows_fixed8
from .rng import LegacyDraws


class HamitonianMC(object):
    def __init__(self, UserDefinedModel):
        if not hasattr(UserDefinedModel, "_engine"):
            raise TypeError("HamitonianMC needs a device-resident model (GravMagModule); "
                            "there is no host implementation of the trajectory")
        self.invert_Mass = None
        self.model = UserDefinedModel
        self.dobs = np.zeros(2)
        self.boundaries = np.zeros((2, 2))
        self.dt = None
        self.Lrange = [10, 50]
        self.seed = None
        self.myrank = None
        self.save_folder = None
        self.cache = {}
        #: where accepted models go: "text" = the reference's model.dat ('%.8f' rows, hmc.py:328-332);
        #: "binary" = float64 rows appended to model.bin (no formatting cost: at 5*10^5 cells one text
        #: row is 5.5 MB and takes longer to format than the trajectory that produced it);
        #: "none" = only the device-side window of the last `posterior_last` models is kept.
        self.sample_sink = "text"
        self.posterior_last = 100
        self._chain_x = None  # identity of the host vector the device chain state mirrors
        #: work of the last sample() call: trajectories run and leapfrog steps taken (accepted or not)
        self.trajectories = 0
        self.leapfrog_steps = 0

    def _kinetic(self, p):
        """Kinetic energy with the (identity) inverse mass matrix (hmc.py:44-50)."""
        return np.dot(self.invert_Mass @ p, p) * 0.5

    def _misfit_and_grad(self, x, alpha):
        """One potential evaluation through the drop-in entry point (hmc.py:71-78)."""
        return self.model.misfit_and_grad(x, self.aprior_model, self.low, self.high,
                                          self.constraint, self.log_factor, alpha,
                                          regulization=self.regularization, beta=self.beta)

    def _kernelw(self):
        return self.model.kernelw()

    # ------------------------------------------------------------------ trajectory
    def _leapfrog(self, xcur, dt, L, alpha, fignum):
        """One trajectory (hmc.py:85-177).  Draws the momentum and the Metropolis variate
        from the global NumPy stream like the reference, runs the L steps on the device.
        Returns (x, U, dsyn, AcceptFlag, U_data, U_model)."""
        n = len(xcur)
        pcur = np.random.randn(n) * self.Sigma
        if self.constraint != 'mandatory':
            return self._leapfrog_unfused(xcur, pcur, dt, L, alpha)
        eng = self.model._engine
        self.model._use_reg(self.regularization, alpha, self.beta, self.aprior_model)
        if self._chain_x is None or self._chain_x is not xcur or not eng._chain_valid:
            eng.chain_init(xcur, self.low, self.high)
        u = np.random.rand()
        accepted, out5 = eng.chain_trajectory(pcur, dt, L, u)
        xnew = eng.chain_get_x() if accepted else xcur
        self._chain_x = xnew
        return xnew, out5[0], _LazyDsyn(eng), accepted, out5[1], out5[2]

    def _leapfrog_unfused(self, xcur, pcur, dt, L, alpha):
        """'logarithmic' constraint (x is not the weighted model): the vector updates stay on
        the host, every potential evaluation is a device call (gh_misfit_and_grad)."""
        pnew = pcur * 1.0
        xnew = xcur * 1.0
        K = self._kinetic(pnew)
        U, grad, dsyn, U_data, U_model = self._misfit_and_grad(xnew, alpha)
        Hcur = K + U
        dsyn_new, Unew, Unew_data, Unew_model = dsyn.copy(), U, U_data, U_model
        pnew -= dt * grad * 0.5
        for i in range(L):
            xnew += dt * pnew
            Unew, grad, dsyn_new, Unew_data, Unew_model = self._misfit_and_grad(xnew, alpha)
            if i < L - 1:
                pnew -= dt * grad
            else:
                pnew -= dt * grad * 0.5
        pnew = -pnew
        Hnew = self._kinetic(pnew) + Unew
        AcceptFlag = False
        u = np.random.rand()
        if Hnew < Hcur or u < np.exp(-(Hnew - Hcur)):
            xcur, U, dsyn, AcceptFlag = xnew, Unew, dsyn_new, True
            U_data, U_model = Unew_data, Unew_model
        return xcur, U, dsyn, AcceptFlag, U_data, U_model

    # ------------------------------------------------------------------ sample files
    def _sink(self, name):
        """The chain's file `name`, appended to.  sample() keeps the files it writes open for the run (an open() per
        row costs more than a trajectory of a small problem on the GPU) and closes them when it returns."""
        fh = getattr(self, "_sink_fh", None)
        if fh is None:
            return open(self.save_folder + "/" + name, "ab"), True
        if name not in fh:
            fh[name] = open(self.save_folder + "/" + name, "ab")
        return fh[name], False

    def _save_models_add(self, x):
        # the bytes np.savetxt(f, x, fmt='%.8f', delimiter=' ') writes (hmc.py:241-245)
        f, own = self._sink("model.dat")
        write_rows_fixed8(f, x)
        if own:
            f.close()

    def _save_misfit_add(self, misfit):
        f, own = self._sink("misfit.dat")
        write_rows_fixed8(f, misfit)
        if own:
            f.close()

    def _to_mw(self, x):
        if self.constraint == 'logarithmic':
            return (self.low + self.high * np.e ** (self.log_factor * x)) / \
                   (1 + np.e ** (self.log_factor * x))
        elif self.constraint == 'mandatory':
            return x
        raise ValueError("Please choose right boundary constraint(mandatory, logarithmic)!")

    def sample(self, nsamples, ndraws, **kwargs):
        """Draw until ndraws+nsamples proposals have been ACCEPTED (hmc.py:252-343)."""
        if not os.path.exists(self.save_folder):
            os.makedirs(self.save_folder)
        if os.path.exists(self.save_folder + "/" + "model" + ".dat"):
            os.remove(self.save_folder + "/" + "model" + ".dat")
        np.random.seed(self.seed)
        _, WmInv, _ = self._kernelw()
        mw = self.initial_model
        print("initial mw:", mw)
        print("mw boundaryies:", self.high, self.low)
        if self.constraint == 'logarithmic':
            x = (1 / self.log_factor) * np.log((mw - self.low) / (self.high - mw))
            print("Using logarithmic boundary constraint.")
        elif self.constraint == 'mandatory':
            x = mw
            print("Using mandatory boundary constraint.")
        else:
            raise ValueError("Please choose right boundary constraint(mandatory, logarithmic)!")
        data_size = self.dobs.shape[0]
        model_size = self.initial_model.shape[0]
        misfit = np.zeros((1, 7))
        m_cache = np.zeros((1, len(x)))
        ncount = 0
        i = 0
        alpha = self.RegulFactor
        self._chain_x = None
        state = {"x": x, "i": 0, "ncount": 0}

        window = self.constraint == 'mandatory' and self.posterior_last > 0
        if window and not getattr(self.model._engine, "_has_window", False):
            self.model._engine.posterior_window(self.posterior_last)
        if self.sample_sink not in ("text", "binary", "none"):
            raise ValueError("sample_sink must be 'text', 'binary' or 'none'")
        if self.sample_sink == "binary" and os.path.exists(self.save_folder + "/model.bin"):
            os.remove(self.save_folder + "/model.bin")

        fused = self.constraint == 'mandatory' and ndraws + nsamples > 0
        # (WmInv is diagonal: hmc.py:328's WmInv @ mw is an element-wise product, the same bits)
        wdiag = WmInv.diagonal() if hasattr(WmInv, "diagonal") and getattr(WmInv, "nnz", -1) == WmInv.shape[0] else None
        unweight = (lambda v: wdiag * v) if wdiag is not None else (lambda v: WmInv @ v)

        def record(U, U_data, U_model, AcceptFlag, get_x):
            """Bookkeeping of one finished trajectory (hmc.py:299-342)."""
            U_data_normed = U_data / data_size
            U_model_normed = U_model / model_size
            U_normed = U_data_normed + alpha * U_model_normed
            if AcceptFlag:
                xs_ = get_x()
                if xs_ is not None:
                    state["x"] = xs_
                if state["i"] >= ndraws:
                    misfit[0, :] = (U, U_data, U_model, U_normed, U_data_normed, U_model_normed,
                                    alpha)
                    self._save_misfit_add(misfit)
                    if self.sample_sink == "text":
                        m = unweight(self._to_mw(state["x"]))
                        m_cache[0, :] = m.copy()
                        self._save_models_add(m_cache)
                    elif self.sample_sink == "binary":
                        f, own = self._sink("model.bin")
                        np.ascontiguousarray(unweight(self._to_mw(state["x"]))).tofile(f)
                        if own:
                            f.close()
                    if window and not fused:
                        self.model._engine.posterior_add()
                state["i"] += 1
            state["ncount"] += 1
            msg = "chain {}: {:.2%}, misfit(total, data, alpha, model)=({:.7f},{:.7f},{:.2f},{:.7f}) " \
                  "-- accept ratio {:.2%}\n". \
                format(self.myrank, state["i"] / (ndraws + nsamples), U_normed, U_data_normed, alpha,
                       U_model_normed, state["i"] / state["ncount"])
            print(msg)
            sys.stdout.flush()
            return state["i"] < ndraws + nsamples

        if self.constraint == 'mandatory' and ndraws + nsamples > 0:
            # device-resident chain, trajectories pipelined (see Engine.run_chain): the random
            # numbers are drawn in the reference's order, only earlier in wall-clock time
            eng = self.model._engine
            self.model._use_reg(self.regularization, alpha, self.beta, self.aprior_model)
            eng.chain_init(x, self.low, self.high)
            n = len(x)

            # the reference's draws per trajectory -- randint, randn(n) * Sigma, rand (hmc.py:297,95,164)
            # -- continue np.random's stream inside the library (inversion/rng.py: same numbers, the
            # normals scaled on several threads) and are handed back to np.random afterwards
            if os.environ.get("GRAVHMC_HOST_RNG", "native") == "numpy":
                def draws():
                    while True:
                        L = np.random.randint(self.Lrange[0], self.Lrange[1] + 1)
                        p0 = np.random.randn(n) * self.Sigma
                        yield L, p0, np.random.rand()
                source = draws()
            else:
                source = LegacyDraws(n, self.Lrange, self.Sigma)
            def on_result(L, acc, o, xs):
                self.trajectories += 1
                self.leapfrog_steps += int(L)
                return record(o[0], o[1], o[2], acc, lambda: xs)

            self.trajectories = self.leapfrog_steps = 0
            self._sink_fh = {}
            try:
                eng.run_chain(source, self.dt, on_result,
                              stop_at_accepts=ndraws + nsamples, record_from=ndraws,
                              want_x=self.sample_sink != "none", overlap=True)
            finally:
                for fh in self._sink_fh.values():
                    fh.close()
                self._sink_fh = None
                if hasattr(source, "release"):
                    source.release()
            self._chain_x = state["x"]
            return state["x"]
        self.trajectories = self.leapfrog_steps = 0
        while state["i"] < ndraws + nsamples:
            L = np.random.randint(self.Lrange[0], self.Lrange[1] + 1)
            self.trajectories += 1
            self.leapfrog_steps += int(L)
            xn, U, _, AcceptFlag, U_data, U_model = self._leapfrog(state["x"], self.dt, L, alpha,
                                                                  state["i"])
            record(U, U_data, U_model, AcceptFlag, lambda xn=xn: xn)
        x = state["x"]
        return x


class _LazyDsyn(object):
    """dsyn of the current chain state, fetched from the device only if somebody reads it
    (the reference's sample loop discards it, hmc.py:299)."""

    def __init__(self, engine):
        self._engine = engine
        self._v = None

    def __array__(self, dtype=None, copy=None):
        if self._v is None:
            self._v = self._engine.chain_get_dsyn()
        return self._v if dtype is None else self._v.astype(dtype)

    def copy(self):
        return np.array(self)


def HMCSample(model, nsamples, ndraws, delta, Lrange,
              initial_model, aprior_model, boundaries, constraint, log_factor, dobs,
              adaptiveRegul, RegulRate, RegulFactor, regularization, beta,
              seed, Sigma, nbest=100, myrank=0, save_folder="mychain",
              plotsamples=False, im=[0, 0], sample_sink="text", posterior_last=100):
    """Set up one chain and run it (hmc.py:358-403).  Chains of different ranks are
    independent: seed + myrank, folder save_folder + str(myrank)."""
    chain = HamitonianMC(model)
    chain.myrank = myrank
    chain.save_folder = save_folder + str(myrank)
    chain.seed = seed + myrank
    chain.nbest = nbest
    nt = boundaries.shape[0]
    chain.boundaries = boundaries
    chain.constraint = constraint
    chain.log_factor = log_factor
    chain.Lrange = Lrange
    chain.dt = delta
    chain.Sigma = Sigma
    chain.adaptiveRegul = adaptiveRegul
    chain.RegulRate = RegulRate
    chain.RegulFactor = RegulFactor
    chain.regularization = regularization
    chain.beta = beta
    row = np.arange(0, nt)
    chain.invert_Mass = coo_matrix((np.ones(nt), (row, row))).tocsr()
    _, _, Wm = chain._kernelw()
    chain.low = Wm @ chain.boundaries[:, 0]
    chain.high = Wm @ chain.boundaries[:, 1]
    chain.im = im
    chain.initial_model = Wm @ initial_model
    chain.aprior_model = Wm @ aprior_model
    chain.dobs = dobs
    chain.plotsamples = plotsamples
    chain.sample_sink = sample_sink
    chain.posterior_last = posterior_last
    chain.sample(nsamples, ndraws)
    return chain


def HMCSampleBatch(model, n_chains, nsamples, ndraws, delta, Lrange,
                   initial_model, aprior_model, boundaries, constraint, log_factor, dobs,
                   adaptiveRegul, RegulRate, RegulFactor, regularization, beta,
                   seed, Sigma, nbest=100, first_rank=0, save_folder="mychain", sample_sink="text"):
    """`n_chains` (<= 16) independent chains on ONE GPU against ONE copy of the kernel matrix.

    The reference runs K chains as K MPI ranks, each rebuilding its own G (run_main.sh:17,
    hmc.py:367-369).  Here chain c is the chain the reference's rank first_rank + c would run --
    its own legacy RandomState(seed + rank) drawn in the reference's order (L, momentum,
    Metropolis variate), its own folder save_folder + str(rank), the same console lines -- but
    every leapfrog step of all chains shares two sweeps of G on the fp64 MFMA path, the chains
    running desynchronised (gh_batch_run: none waits for the longest trajectory of a round); on
    problems small enough they take turns inside the resident chain kernel instead.  Chains that
    reach ndraws + nsamples accepted samples keep running (unrecorded) until the slowest one is
    done.  'mandatory' constraint only."""
    if constraint != 'mandatory':
        raise ValueError("HMCSampleBatch supports the 'mandatory' boundary constraint only")
    eng = model._engine
    _, WmInv, Wm = model.kernelw()
    low, high = Wm @ boundaries[:, 0], Wm @ boundaries[:, 1]
    mw0, mwapr = Wm @ initial_model, Wm @ aprior_model
    model._use_reg(regularization, RegulFactor, beta, mwapr)
    M, N = mw0.shape[0], np.asarray(dobs).shape[0]
    ranks = [first_rank + c for c in range(n_chains)]
    # one legacy stream per chain, bit for bit np.random.RandomState(seed + rank), drawn by the library (a
    # trajectory's 6000 normals cost NumPy 57 us on a GPU box's core -- with 16 chains in lock-step the GPU
    # needs 23 us for a step of ALL chains; the native draws take half of NumPy's time and run one thread per chain)
    from .rng import LegacyDraws, draw_workers
    # Trajectories offered per chain and library call.  The chains run desynchronised
    # (gh_batch_run, carry-over mode): a call ends when the first chain has used up its offer, the
    # others keep their trajectory in flight; what a chain has not started is offered again next
    # time together with fresh draws.  Bounded by the momenta held on the host.
    # (up to 32 per call: at 6000 cells a lock-step of 16 chains takes 23 us on the GPU -- 8 trajectories per chain
    # are 2 ms of kernel against ~1.5 ms of staging, copies and Python per call)
    T = int(max(2, min(32, (256 << 20) // (8 * M * n_chains))))
    # (each chain draws into a ring of 2 T page-locked rows: the library sends them to the GPU from where they lie)
    n_draw = draw_workers(n_chains)
    rs = [LegacyDraws(M, Lrange, Sigma, seed=seed + r, helpers=1 if n_draw >= 8 else None).use_ring(eng, 2 * T) for r in ranks]
    folders = [save_folder + str(r) for r in ranks]
    for f in folders:
        os.makedirs(f, exist_ok=True)
        for name in ("model.dat", "model.bin"):
            if os.path.exists(f + "/" + name):
                os.remove(f + "/" + name)
    print("initial mw:", mw0)
    print("mw boundaryies:", high, low)
    print("Using mandatory boundary constraint.")
    eng.batch_init(np.stack([mw0] * n_chains), low, high)
    alpha = RegulFactor
    acc_n = [0] * n_chains
    tot_n = [0] * n_chains
    target = ndraws + nsamples
    from concurrent.futures import ThreadPoolExecutor
    pool = ThreadPoolExecutor(max_workers=2)
    # (a draw is mostly sequential: one generator per core, rng.draw_workers)
    draw_pool = ThreadPoolExecutor(max_workers=n_draw)

    import collections
    import queue
    import threading
    pending = [collections.deque() for _ in range(n_chains)]    # drawn, not started yet: (L, row address, u, row)
    inflight = [0] * n_chains                                   # started, result not reported yet

    def top_up():
        """Fill every chain's queue up to 2 T draws (L, momentum, Metropolis variate), each chain
        from its own stream in the reference's order; the streams are independent, so they are
        drawn concurrently."""
        def one(c):
            n = 2 * T - len(pending[c])
            if n > 0:
                pending[c].extend(rs[c].take_ring(n))
        list(draw_pool.map(one, range(n_chains)))

    def offer():
        cur = [[pending[c][t] for t in range(T)] for c in range(n_chains)]
        return ([[tr[1] for tr in ch] for ch in cur], [[tr[0] for tr in ch] for ch in cur],
                [[tr[2] for tr in ch] for ch in cur])

    want_x = sample_sink != "none"
    top_up()
    # (the chains' files stay open for the run: sixteen chains in lock-step deliver ~35 000 results a second at 6000
    # cells, an open() per row would cost more than the GPU; rows are flushed with every call's console lines)
    sink_name = {"text": "model.dat", "binary": "model.bin"}.get(sample_sink)
    f_misfit = [open(f + "/misfit.dat", "ab") for f in folders]
    f_model = [open(f + "/" + sink_name, "ab") for f in folders] if sink_name else None
    wdiag = WmInv.diagonal()

    def report(accepted, out5, xs, n_done):
        """A call's results in the order the reference's ranks would report them: files and console lines
        (hmc.py:318-336 per chain)."""
        lines = []
        for i in range(int(max(n_done))):
            for c in range(n_chains):
                if i >= n_done[c] or acc_n[c] >= target:
                    continue
                U, U_data, U_model = out5[c, i, 0], out5[c, i, 1], out5[c, i, 2]
                Udn, Umn = U_data / N, U_model / M
                Un = Udn + alpha * Umn
                if accepted[c, i]:
                    if acc_n[c] >= ndraws:
                        write_rows_fixed8(f_misfit[c], np.array([[U, U_data, U_model, Un, Udn, Umn, alpha]]))
                        if want_x:
                            m = wdiag * xs[c, i]          # (WmInv is diagonal: hmc.py:328's WmInv @ mw)
                            if sample_sink == "text":
                                write_rows_fixed8(f_model[c], m[None, :])
                            else:
                                m.tofile(f_model[c])
                    acc_n[c] += 1
                tot_n[c] += 1
                lines.append("chain {}: {:.2%}, misfit(total, data, alpha, model)=({:.7f},{:.7f},{:.2f},{:.7f}) "
                             "-- accept ratio {:.2%}\n\n".format(ranks[c], acc_n[c] / target, Un, Udn, alpha, Umn,
                                                               acc_n[c] / tot_n[c]))
        sys.stdout.write("".join(lines))
        sys.stdout.flush()
        for fh in f_misfit + (f_model or []):
            fh.flush()

    # The reporting runs on a thread of its own BESIDE the next library call (which waits for the GPU with the
    # interpreter lock released): at 6000 cells a call's ~500 results cost Python as long as the GPU needs for them.
    # The loop itself only counts accepted samples to know when every chain has its target.
    results = queue.Queue(maxsize=4)
    failure = []

    def reporter():
        while True:
            item = results.get()
            if item is None:
                return
            if failure:
                continue                                  # (keep draining: the producer must not block)
            try:
                report(*item)
            except BaseException as e:                    # noqa: B902 -- handed to the caller below
                failure.append(e)

    rep = threading.Thread(target=reporter, name="hmc-batch-report", daemon=True)
    rep.start()
    acc_fast = np.zeros(n_chains, dtype=np.int64)
    try:
        while acc_fast.min() < target and not failure:
            p0s, Ls, us = offer()
            # more is drawn on the host while the GPU runs (appended behind what has been offered)
            entered = threading.Event()
            fut = pool.submit(eng.batch_run, p0s, delta, Ls, us, want_x, True, entered)
            # (the GPU has its work before the drawing threads compete for the interpreter lock)
            while not entered.wait(0.05):
                if fut.done():
                    break
            top_up()
            accepted, out5, xs, n_started, n_done = fut.result()
            for c in range(n_chains):
                for _ in range(int(n_started[c])):
                    pending[c].popleft()
                inflight[c] += int(n_started[c]) - int(n_done[c])
                acc_fast[c] += int(accepted[c, :int(n_done[c])].sum())
            results.put((accepted, out5, xs, n_done))
    finally:
        results.put(None)
        rep.join()
        for fh in f_misfit + (f_model or []):
            fh.close()
        pool.shutdown(wait=False)
        draw_pool.shutdown(wait=False)
        for r in rs:
            r.release()
    if failure:
        raise failure[0]
    return acc_n, tot_n
