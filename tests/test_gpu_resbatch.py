"""Chains in LOCK-STEP inside the resident batch kernel (csrc/resbatch.hip.h; BASELINE configs[0] -- north_star's
target configuration -- and configs[2] with several chains per GPU; the reference runs its chains as MPI ranks,
inversion/hmc.py:367-369).  Every chain is checked against the ORACLE's trajectories
(`oracle.Problem.leapfrog`: inversion/hmc.py:85-177 over inversion/potential.py:688-845) on the oracle's own
kernel matrix, decisions included -- not against another HIP kernel."""
import numpy as np
import pytest

from helpers import c1_inputs, relmax
from conftest import gold

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(built_lib):
    import gravinv3dhmc_amd as g
    return g


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


@pytest.fixture(scope="module")
def c1(orc):
    """C1's kernel from the oracle (600 x 6000 prism entries), weighted, with noisy synthetic data."""
    mesh, xp, yp, zp = c1_inputs()
    K = orc.prism_gz_kernel(xp, yp, zp, mesh.cell_bounds())
    Aw, wm = orc.col_weight(K)
    rho = np.zeros(mesh.shape)
    rho[2:5, 10:18, 7:11] = 1.0
    d = K @ rho.ravel()
    dobs = d + 0.02 * np.abs(d).max() * np.random.default_rng(0).normal(size=d.size)
    return mesh, (xp, yp, zp), Aw, wm, dobs


def _engine(G, c1, reg, alpha, beta):
    mesh, obs, Aw, wm, dobs = c1
    e = G.Engine(obs[0].size, mesh.size)
    e.set_obs(*obs)
    e.set_cells(mesh.cell_bounds(), 0)
    e.build_G()
    w = e.weight(0.5)
    assert relmax(w, wm) < 1e-11
    e.set_data(dobs)
    e.set_reg(reg, alpha, beta, mesh.shape, 0.001 * wm)
    return e


class _Lists(object):
    """Per chain: the trajectories it has been offered (L, p0, u) and the oracle's chain through them."""

    def __init__(self, P, x0s, low, high, dt, rng, sig, Lmax):
        self.P, self.low, self.high, self.dt, self.rng, self.sig, self.Lmax = P, low, high, dt, rng, sig, Lmax
        self.C, self.M = x0s.shape
        self.xo = [x.copy() for x in x0s]
        self.pending = [[] for _ in range(self.C)]      # drawn, not yet reported as started
        self.started = [[] for _ in range(self.C)]      # started, result not yet checked
        self.n_acc = self.n_rej = self.n = 0
        self.worst = 0.0

    def offer(self, T):
        for c in range(self.C):
            while len(self.pending[c]) < T:
                L = int(self.rng.integers(1, self.Lmax + 1))
                p0 = self.rng.normal(size=self.M) * self.sig * (1.0 + 0.5 * c)
                u = float(self.rng.uniform()) * (0.02 if self.rng.uniform() < 0.3 else 1.0)
                self.pending[c].append((L, p0, u))
        return ([[t[1] for t in self.pending[c][:T]] for c in range(self.C)],
                [[t[0] for t in self.pending[c][:T]] for c in range(self.C)],
                [[t[2] for t in self.pending[c][:T]] for c in range(self.C)])

    def report(self, acc, out5, xs, n_started, n_done):
        for c in range(self.C):
            for _ in range(int(n_started[c])):
                self.started[c].append(self.pending[c].pop(0))
            for i in range(int(n_done[c])):
                L, p0, u = self.started[c].pop(0)
                self.xo[c], ao, oo, _ = self.P.leapfrog(self.xo[c], p0, self.dt, L, self.low, self.high, u)
                assert bool(acc[c, i]) == ao, (c, i, out5[c, i], oo)
                self.worst = max(self.worst, relmax(out5[c, i], oo))
                if ao and xs is not None:
                    self.worst = max(self.worst, relmax(xs[c, i], self.xo[c]))
                self.n_acc += ao
                self.n_rej += not ao
                self.n += 1


@pytest.mark.parametrize("reg,C", [("Damping", 16), ("MS", 5), ("TV", 16), ("Smoothness", 3)])
def test_lockstep_chains_against_oracle_trajectories(G, orc, c1, reg, C):
    """gh_batch_run in carry-over mode as HMCSampleBatch drives it (offers of T trajectories, a call ends when
    a chain has nothing left to start, the others stay in flight), then a drain (T = 0): every result of every
    chain against the oracle, <= 1e-10, identical decisions, accepted states included."""
    mesh, obs, Aw, wm, dobs = c1
    alpha, beta = (1.0, 0.001) if reg in ("TV", "MS") else (1.0, 0.01)
    P = orc.Problem(Aw, dobs, 0.001 * wm, reg, alpha, beta, wm=wm, shape=mesh.shape)
    e = _engine(G, c1, reg, alpha, beta)
    low, high = 0.0 * wm, 1.0 * wm
    rng = np.random.default_rng(5)
    x0s = np.stack([(0.001 + 0.01 * c) * wm for c in range(C)])
    dt = {"MS": 0.02, "Smoothness": 0.055}.get(reg, 0.045)   # (step sizes at which part of the proposals is rejected)
    e.batch_init(x0s, low, high)
    lists = _Lists(P, x0s, low, high, dt, rng, 0.001, 8)
    T, calls = 3, 0
    while lists.n < 7 * C:
        p0s, Ls, us = lists.offer(T)
        acc, out5, xs, ns, nd = e.batch_run(p0s, dt, Ls, us, want_x=True, carry=True)
        assert max(ns) == T                      # (the call ended because a chain ran out of offers)
        lists.report(acc, out5, xs, ns, nd)
        calls += 1
    acc, out5, xs, ns, nd = e.batch_run([[] for _ in range(C)], dt, np.zeros((C, 0)), np.zeros((C, 0)), want_x=True,
                                        carry=True)
    lists.report(acc, out5, xs, ns, nd)
    assert all(len(s) == 0 for s in lists.started)           # nothing left in flight after the drain
    for c in range(C):
        assert relmax(e.batch_get_x(c), lists.xo[c]) < 1e-10
    st = e.batch_resident_stats()
    print("lock-step chains [%s, %d chains] vs ORACLE trajectories: %d results in %d calls, worst %.2e, accepted %d, "
          "rejected %d; %r" % (reg, C, lists.n, calls, lists.worst, lists.n_acc, lists.n_rej, st))
    assert lists.worst < 1e-10 and lists.n_acc > 0 and lists.n_rej > 0
    assert st["launches"] in (calls, calls + 1) and st["timeouts"] == 0 and st["chain_steps"] > 0
    n_l = st["launches"]
    # complete rounds (gh_batch_trajectory, and gh_batch_run without carry-over) continue the same chains
    Ls = rng.integers(1, 6, size=C)
    p0s = rng.normal(size=(C, mesh.size)) * 0.001
    us = rng.uniform(size=C)
    acc, out5 = e.batch_trajectory(p0s, dt, Ls, us)
    for c in range(C):
        lists.xo[c], ao, oo, _ = P.leapfrog(lists.xo[c], p0s[c], dt, int(Ls[c]), low, high, float(us[c]))
        assert bool(acc[c]) == ao and relmax(out5[c], oo) < 1e-10
    Ls = rng.integers(1, 6, size=(C, 2))
    p0s = rng.normal(size=(C, 2, mesh.size)) * 0.001
    us = rng.uniform(size=(C, 2))
    acc, out5, xs = e.batch_run(p0s, dt, Ls, us, want_x=True)
    for c in range(C):
        for t in range(2):
            lists.xo[c], ao, oo, _ = P.leapfrog(lists.xo[c], p0s[c, t], dt, int(Ls[c, t]), low, high, float(us[c, t]))
            assert bool(acc[c, t]) == ao and relmax(out5[c, t], oo) < 1e-10
            if ao:
                assert relmax(xs[c, t], lists.xo[c]) < 1e-10
    assert e.batch_resident_stats()["launches"] == n_l + 2
    e.close()


def test_lockstep_kernel_gives_up_cleanly_with_trajectories_in_flight(G, orc, c1, monkeypatch):
    """A launch whose workgroups wait for partners that never come (test hook) times out (2 s), changes nothing,
    and the call continues on the chains-take-turns kernel: the trajectories that were in flight are replayed
    there from their chains' current samples with their own momentum, length and variate (kept on the device),
    in front of the new lists.  Same chains as the oracle's."""
    mesh, obs, Aw, wm, dobs = c1
    C = 6
    P = orc.Problem(Aw, dobs, 0.001 * wm, "Damping", 1.0, 0.01, wm=wm, shape=mesh.shape)
    e = _engine(G, c1, "Damping", 1.0, 0.01)
    low, high = 0.0 * wm, 1.0 * wm
    x0s = np.stack([(0.001 + 0.01 * c) * wm for c in range(C)])
    e.batch_init(x0s, low, high)
    lists = _Lists(P, x0s, low, high, 0.01, np.random.default_rng(9), 0.001, 8)
    for call in range(4):
        if call == 2:
            monkeypatch.setenv("GRAVHMC_RESBATCH_TEST_ABORT", "1")
        p0s, Ls, us = lists.offer(2)
        acc, out5, xs, ns, nd = e.batch_run(p0s, 0.01, Ls, us, want_x=True, carry=True)
        monkeypatch.delenv("GRAVHMC_RESBATCH_TEST_ABORT", raising=False)
        lists.report(acc, out5, xs, ns, nd)
        if call == 1:
            assert any(len(s) > 0 for s in lists.started)    # (something is in flight when the next launch gives up)
    st = e.batch_resident_stats()
    assert st["timeouts"] == 1 and st["launches"] == 2, st
    assert all(len(s) == 0 for s in lists.started)           # the take-turns kernel leaves nothing in flight
    assert lists.worst < 1e-10 and lists.n >= 4 * C
    for c in range(C):
        assert relmax(e.batch_get_x(c), lists.xo[c]) < 1e-10
    e.close()


def test_lockstep_chains_with_the_compressed_forward_as_baseline_config_3(G, orc):
    """BASELINE configs[2] (segmentgrid, wavelet='3D' compressed forward with the exact dense adjoint, TV:
    potential.py:693-708) with 8 chains in lock-step: LDS holds the dense model-space form of the compressed
    operator, the adjoint's register operand is Aw.  Against oracle.Problem(csr=..., dwt=...)."""
    from oracle import wavelet as ow
    ex = gold("example_inputs.npz")
    obs = ex["seg_obs"]
    M, shape = 6000, (10, 30, 20)
    gm = G.GravMagModule(obs[:, 3], (0, 2000, 0, 3000, 0, 2100), ([100, 200, 300], 100, 100),
                         (obs[:, 0], obs[:, 1], obs[:, 2]), mseg=True, mdivisionsection=[0, 300, 900, 2100],
                         wavelet='3D', verbose=False)
    wm = gm.Wm.diagonal()
    Aw = np.asarray(gm.Aw)
    mwapr, low, high = 0.001 * wm, 0.0 * wm, 1.0 * wm
    P = orc.Problem(Aw, obs[:, 3], mwapr, "TV", 1.0, 0.001, wm=wm, shape=shape, csr=ow.compress_kernel(Aw, 3, shape),
                    dwt=lambda v: ow.model_coeffs(v, 3, shape))
    gm._use_reg("TV", 1.0, 0.001, mwapr)
    e = gm._engine
    C, T, dt = 8, 3, 0.01
    rng = np.random.default_rng(3)
    x0s = np.stack([(0.001 + 0.02 * c) * wm for c in range(C)])
    e.batch_init(x0s, low, high)
    Ls = rng.integers(2, 9, size=(C, T))
    p0s = rng.normal(size=(C, T, M)) * 0.001
    us = rng.uniform(size=(C, T))
    acc, out5, xs = e.batch_run(p0s, dt, Ls, us, want_x=True)
    worst = 0.0
    for c in range(C):
        xo = x0s[c]
        for t in range(T):
            xo, ao, oo, _ = P.leapfrog(xo, p0s[c, t], dt, int(Ls[c, t]), low, high, float(us[c, t]))
            assert bool(acc[c, t]) == ao, (c, t, out5[c, t], oo)
            worst = max(worst, relmax(out5[c, t], oo))
            if ao:
                worst = max(worst, relmax(xs[c, t], xo))
    st = e.batch_resident_stats()
    print("C3 (wavelet 3D forward, TV), 8 chains in lock-step vs the ORACLE: worst %.2e; %r" % (worst, st))
    assert worst < 1e-9 and st["launches"] == 1 and st["timeouts"] == 0
    e.close()


def test_momentum_rows_in_pinned_memory_go_straight_to_the_device(G, c1):
    """gh_pinned_alloc / LegacyDraws.use_ring: rows of a page-locked ring are sent from where they lie (adjacent
    rows of a chain in one copy, the ring's wrap-around splits a run), rows anywhere else are gathered first --
    and a list mixing both kinds is still the same list.  Same draws (RandomState(seed + chain): hmc.py:369,
    91, 164) three ways: identical decisions, potentials and samples, bit for bit."""
    from gravinv3dhmc_amd.inversion.rng import LegacyDraws
    mesh, obs, Aw, wm, dobs = c1
    C, T, M = 4, 6, mesh.size
    low, high = 0.0 * wm, 1.0 * wm
    x0s = np.stack([0.001 * wm * (1.0 + 0.1 * k) for k in range(C)])
    outs = []
    for mode in ("arrays", "ring", "mixed"):
        e = _engine(G, c1, "Damping", 1e-3, 0.0)
        e.batch_init(x0s, low, high)
        draws = [LegacyDraws(M, (3, 9), 0.001, seed=50 + k) for k in range(C)]
        if mode != "arrays":
            for d in draws:
                d.use_ring(e, T + 2)
                d.take_ring(4)                       # (moves the head: the next 6 rows wrap around the ring's end)
        else:
            for d in draws:
                d.take_block(4)
        p0s, Ls, us = [], [], []
        for k, d in enumerate(draws):
            if mode == "arrays":
                Lk, pk, uk = d.take_block(T)
                rows = [pk[i] for i in range(T)]
            else:
                got = d.take_ring(T)
                Lk, uk = [g[0] for g in got], [g[2] for g in got]
                rows = [g[1] for g in got]
                if mode == "mixed":                  # every other row from ordinary memory, as arrays
                    rows = [d.ring_row(g[3]).copy() if i % 2 else d.ring_row(g[3]) for i, g in enumerate(got)]
            p0s.append(rows)
            Ls.append(list(Lk))
            us.append(list(uk))
        acc, out5, xs = e.batch_run(p0s, 0.05, Ls, us, want_x=True)
        st = e.batch_staging_stats()
        if mode == "arrays":
            assert st["rows_direct"] == 0 and st["rows_staged"] == C * T
        elif mode == "ring":
            assert st["rows_direct"] == C * T and st["rows_staged"] == 0
        else:
            assert st["rows_direct"] == C * T // 2 and st["rows_staged"] == C * T // 2
        assert e.batch_resident_stats()["launches"] >= 1
        outs.append((acc, out5, np.where(acc[:, :, None], xs, 0.0), np.stack([e.batch_get_x(k) for k in range(C)])))
        for d in draws:
            d.release()
        e.close()
    assert outs[0][0].any()
    for o in outs[1:]:
        for a, b in zip(outs[0], o):
            assert np.array_equal(a, b)
