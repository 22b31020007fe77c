"""Round-3 kernel families pinned to the ORACLE (oracle/gravhmc_oracle.c: the C restatement of
gravmag/_prism.pyx:265-290, gravmag/_tesseroid_numba.py:32-157,207-222, inversion/potential.py:688-845 and
inversion/hmc.py:85-177), not to other HIP kernels: the matrix-free batch passes (`mfb_*`, both forms), the
single chain on teams (`mf_team_kernel`), the shift-invariant store (`lonsym_sweep_kernel`), the stored-kernel
batch on teams (`batch_team_kernel`) at the C2 shape, and oracle columns of G at the full sizes of C2 and of
the C5 share."""
import numpy as np
import pytest

from helpers import relmax
from test_gpu_mfbatch import CASES, _global_model

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(built_lib):
    import gravinv3dhmc_amd as g
    return g


@pytest.fixture(scope="module")
def orc():
    from oracle import oracle
    return oracle


def _oracle_kernel(orc, obs, bounds, kind):
    if kind == 1:
        return orc.tess_gz_kernel(obs[0], obs[1], obs[2], bounds)
    return orc.prism_gz_kernel(obs[0], obs[1], obs[2], bounds)


@pytest.mark.parametrize("form", ["teams_one_evaluation", "two_passes"])
@pytest.mark.parametrize("case", list(CASES))
def test_matrix_free_batch_and_team_chain_against_oracle_trajectories(G, orc, monkeypatch, case, form):
    """Every chain of a matrix-free batch (mfb_fused_kernel on teams / mfb_adjoint + mfb_forward) and the
    single matrix-free chain (mf_team_kernel where the near-field list exists, mf_fused_kernel otherwise)
    against `oracle.Problem.leapfrog` on the oracle's own kernel matrix: potentials and Hamiltonians
    <= 1e-10, positions <= 1e-10, identical Metropolis decisions."""
    problem, exact, near, reg = CASES[case]
    monkeypatch.setenv("GRAVHMC_MF_EXACT", exact)
    monkeypatch.setenv("GRAVHMC_MF_NEAR", near)
    monkeypatch.setenv("GRAVHMC_MFB_FUSED", "0" if form == "two_passes" else "1")
    obs, bounds, kind, shape = problem(G)
    N, M = obs[0].size, bounds.shape[0]
    rng = np.random.default_rng(7)
    rho = rng.uniform(0.0, 0.5, M)
    K = _oracle_kernel(orc, obs, bounds, kind)
    Aw, wm = orc.col_weight(K)
    d_true = K @ rho
    dobs = d_true + 0.02 * np.abs(d_true).max() * rng.normal(size=N)
    alpha, beta = 0.05, 0.01
    P = orc.Problem(Aw, dobs, 0.001 * wm, reg, alpha, beta, wm=wm, shape=shape)

    def make():
        e = G.Engine(N, M)
        e.set_matrix_free(True)
        e.set_obs(*obs)
        e.set_cells(bounds, kind, 1.6)
        e.build_G()
        w = e.weight(0.5)
        assert relmax(w, wm) < 1e-11
        e.set_data(dobs)
        e.set_reg(reg, alpha, beta, shape, 0.001 * wm)
        return e

    eb, es = make(), make()
    assert relmax(eb.forward(wm * rho), d_true) < 1e-10      # (the engine is weighted: Aw (wm rho) = K rho)
    C, dt, sig = 4, 0.005, 0.002
    low, high = 0.0 * wm, 0.8 * wm
    x0s = np.stack([(0.001 + 0.05 * c) * wm for c in range(C)])
    eb.batch_init(x0s, low, high)
    es.chain_init(x0s[1], low, high)
    xo = [x0s[c].copy() for c in range(C)]
    worst_b = worst_s = 0.0
    n_acc = n_rej = 0
    for it in range(3):
        Ls = rng.integers(1, 6, size=C)
        p0s = rng.normal(size=(C, M)) * sig
        us = rng.uniform(size=C) * (0.05 if it == 1 else 1.0)
        acc, out5 = eb.batch_trajectory(p0s, dt, Ls, us)
        a1, o1 = es.chain_trajectory(p0s[1], dt, int(Ls[1]), float(us[1]))
        for c in range(C):
            xo[c], ao, oo, _ = P.leapfrog(xo[c], p0s[c], dt, int(Ls[c]), low, high, float(us[c]))
            assert bool(acc[c]) == ao, (case, form, it, c, out5[c], oo)
            worst_b = max(worst_b, relmax(out5[c], oo), relmax(eb.batch_get_x(c), xo[c]))
            n_acc += ao
            n_rej += not ao
            if c == 1:      # the single chain runs the oracle's chain 1
                assert bool(a1) == ao, (case, form, it, o1, oo)
                worst_s = max(worst_s, relmax(o1, oo), relmax(es.chain_get_x(), xo[1]))
    fs, ts = eb.batch_fused_stats(), es.matrix_free_team_stats()
    print("matrix-free [%s, %s] vs ORACLE trajectories: batch of %d chains worst %.2e (accepted %d, rejected %d), "
          "single chain %.2e; batch team form %r, single-chain team form %r"
          % (case, form, C, worst_b, n_acc, n_rej, worst_s, fs, ts))
    assert worst_b < 1e-10 and worst_s < 1e-10 and n_acc > 0
    if form == "teams_one_evaluation":
        assert fs["launches"] > 0 and fs["timeouts"] == 0
    if kind == 1 and near == "1":
        assert ts["launches"] > 0 and ts["timeouts"] == 0      # mf_team_kernel ran the single chain
    for e in (eb, es):
        e.close()


@pytest.mark.parametrize("case", ["coarse_odd_sizes", "c4_full_size", "coarse_odd_sizes_streamed", "c4_full_size_streamed"])
def test_shift_invariant_store_against_the_oracle(G, orc, case, monkeypatch):
    """lonsym_sweep_kernel's table against the oracle's tesseroid kernel itself.  coarse: the whole dense K
    (forward, weights, adjoint, potential + gradient for a cell-local and a stencil regulariser, a chain
    against oracle.Problem.leapfrog).  C4 at full size: 68 oracle rows x 72000 cells (64 random
    observations, two polar rows, two more) against K[i, :] = wm * (Aw^T e_i) read through the table."""
    rng = np.random.default_rng(13)
    if case.endswith("_streamed"):      # the harmonic form as streaming passes over T^ (lonsymw.hip.h), forced
        monkeypatch.setenv("GRAVHMC_LONSYM_WIDE", "2")
    if case.startswith("coarse_odd_sizes"):
        mesh, lon, lat, h = _global_model(G, 10.0, 15.0, -1000000, 30000.0)
        h[::3] = 45000.0
        perm = rng.permutation(lon.size)
        lon, lat, h = lon[perm], lat[perm], h[perm]
    else:
        mesh, lon, lat, h = _global_model(G, 3.0, 3.0, -300000, 5000.0)
    N, M = lon.size, mesh.size
    bounds = mesh.cell_bounds()
    t = G.Engine(N, M)
    t.set_shift_invariant(True)
    t.set_obs(lon, lat, h)
    t.set_cells(bounds, 1, 1.6)
    t.build_G()
    assert t.shift_invariant_info()["n_lon"] == mesh.shape[2]
    assert t.shift_invariant_harmonic()["form"] == ("streamed" if case.endswith("_streamed") else "registers")
    if case.startswith("c4_full_size"):
        wt = t.weight(0.5)
        rows = np.r_[rng.choice(N, 64, replace=False), [0, 60, 60 * 61 + 30, N - 1]]
        Ko = orc.tess_gz_kernel(lon[rows], lat[rows], h[rows], bounds)
        e_row = 0.0
        for q, i in enumerate(rows):
            ei = np.zeros(N)
            ei[i] = 1.0
            Ki = t.adjoint(ei) * wt
            e_row = max(e_row, float(np.abs(Ki - Ko[q]).max() / np.abs(Ko[q]).max()))
            assert (np.abs(Ki - Ko[q]) / np.maximum(np.abs(Ko[q]), 1e-300)).max() < 1e-9, i
        cols = np.r_[0, 1, 119, 36000, 71880, 71999]
        Kc = orc.tess_gz_kernel(lon, lat, h, bounds[cols])
        e_w = relmax(wt[cols], np.sqrt((Kc ** 2).sum(0)))
        print("shift-invariant store at C4 size vs the ORACLE: 68 rows x 72000 cells %.2e of a row's largest entry, "
              "6 column norms %.2e" % (e_row, e_w))
        assert e_row < 1e-10 and e_w < 1e-11
        t.close()
        return
    K = orc.tess_gz_kernel(lon, lat, h, bounds)
    rho = rng.uniform(0.0, 0.5, M)
    d_true = K @ rho
    e_fwd = relmax(t.forward(rho), d_true)
    Aw, wm = orc.col_weight(K)
    e_w = relmax(t.weight(0.5), wm)
    r = rng.normal(size=N)
    e_adj = relmax(t.adjoint(r), Aw.T @ r)
    dobs = d_true + 0.02 * np.abs(d_true).max() * rng.normal(size=N)
    x = rng.uniform(0, 0.8, M) * wm
    worst = 0.0
    for reg in ("Damping", "TV"):
        t.set_data(dobs)
        t.set_reg(reg, 0.05, 0.01, mesh.shape, 0.001 * wm)
        P = orc.Problem(Aw, dobs, 0.001 * wm, reg, 0.05, 0.01, wm=wm, shape=mesh.shape)
        a, b = t.misfit_and_grad(x), P.misfit_and_grad(x)
        worst = max(worst, abs(a[0] - b[0]) / abs(b[0]), relmax(a[1], b[1]), relmax(a[2], b[2]))
    print("shift-invariant store [coarse] vs the ORACLE's dense kernel: forward %.2e weights %.2e adjoint %.2e "
          "potential/gradient %.2e" % (e_fwd, e_w, e_adj, worst))
    assert e_fwd < 1e-10 and e_w < 1e-11 and e_adj < 1e-10 and worst < 1e-10
    low, high = 0.0 * wm, 0.8 * wm
    t.chain_init(0.001 * wm, low, high)
    xo = 0.001 * wm
    for _ in range(5):
        L, p0, u = int(rng.integers(2, 7)), rng.normal(size=M) * 0.001, float(rng.uniform())
        a1, o1 = t.chain_trajectory(p0, 0.005, L, u)
        xo, ao, oo, _ = P.leapfrog(xo, p0, 0.005, L, low, high, u)
        assert bool(a1) == ao and relmax(o1, oo) < 1e-10
    assert relmax(t.chain_get_x(), xo) < 1e-10
    t.close()


def _prism_columns_against_oracle(eng, orc, xp, yp, zp, bounds, cols, tag):
    """forward(e_j) of the unweighted stored kernel = column j of G, against the oracle's entries; then
    the weights against the oracle's column norms.  Two measures: the entry error against the largest
    entry of the kernel (what a forward product G rho sees; bound 1e-11) and against the largest entry of
    the column itself (bound 1e-8: a cell 50 cell sizes deep seen from 10 km away is the sum of 24 terms of
    magnitude x log(y + r) ~ 1e5 that cancel to ~1e-2 before the factor G * SI2MGAL, _prism.pyx:265-290 --
    the one- or two-ulp differences between the device's log / atan2 and glibc's are amplified by that
    cancellation; the reference's own entries carry the same absolute uncertainty; measured: 4e-13 / 1.4e-9 at
    C2's 10 km, 1.1e-12 / 3.3e-11 at the 20 km of the C5 share)."""
    Ko = orc.prism_gz_kernel(xp, yp, zp, bounds[cols])
    kmax = float(np.abs(Ko).max())
    worst_abs = worst_col = 0.0
    for q, j in enumerate(cols):
        e = np.zeros(eng.M)
        e[j] = 1.0
        d = eng.forward(e)
        worst_abs = max(worst_abs, float(np.abs(d - Ko[:, q]).max()) / kmax)
        worst_col = max(worst_col, relmax(d, Ko[:, q]))
    wm = eng.weight(0.5)
    e_w = relmax(wm[cols], np.sqrt((Ko ** 2).sum(0)))
    print("%s: %d columns of G vs the ORACLE (%d x %d entries): %.2e of the largest entry, %.2e of the own "
          "column's largest entry; their norms %.2e" % (tag, len(cols), Ko.shape[0], len(cols), worst_abs, worst_col, e_w))
    assert worst_abs < 1e-11 and worst_col < 1e-8 and e_w < 1e-11
    return wm


def test_c2_full_size_oracle_columns_and_team_batch_of_16_chains(G, orc):
    """BASELINE configs[1] at full size (10^4 x 5*10^5, 40 GB): eight columns of G (first, middle, last cells)
    against the oracle's prism entries -- the headline workload touches oracle values -- and
    batch_team_kernel at exactly that shape (teams of 23 workgroups x 11 ranges of column tiles, 16 chains:
    the library's default there): chains 3 and 12 re-run on the single-chain sweep path, identical
    decisions, <= 1e-10."""
    nx = ny = 100
    nz = 50
    mesh = G.mesher.PrismMesh((0, 100.0 * nx, 0, 100.0 * ny, 0, 100.0 * nz), (100, 100, 100))
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 100.0 * ny, ny), np.linspace(0, 100.0 * nx, nx))]
    zp = np.zeros_like(xp)
    N, M = xp.size, mesh.size
    bounds = mesh.cell_bounds()
    rng = np.random.default_rng(21)

    def make():
        e = G.Engine(N, M)
        e.set_obs(xp, yp, zp)
        e.set_cells(bounds, 0)
        e.build_G()
        return e

    eb = make()
    cols = np.r_[0, 1, 2, M // 2, M // 2 + 1, M - 3, M - 2, M - 1]
    wm = _prism_columns_against_oracle(eb, orc, xp, yp, zp, bounds, cols, "C2 full size")
    es = make()
    assert np.array_equal(es.weight(0.5), wm)
    rho = np.zeros(mesh.shape)
    rho[10:25, 40:60, 40:60] = 1.0
    d_true = eb.forward(wm * rho.ravel())
    dobs = d_true + 0.02 * np.abs(d_true).max() * rng.normal(size=N)
    for e in (eb, es):
        e.set_data(dobs)
        e.set_reg("Damping", 1.0, 0.01, mesh.shape, 0.001 * wm)
    C, dt = 16, 0.002
    low, high = 0.0 * wm, 1.0 * wm
    x0s = np.stack([(0.001 + 0.002 * c) * wm for c in range(C)])
    eb.batch_init(x0s, low, high)
    rounds = []
    for it in range(2):
        Ls = rng.integers(2, 5, size=C)
        p0s = rng.normal(size=(C, M)) * 0.001
        us = rng.uniform(size=C)
        acc, out5 = eb.batch_trajectory(p0s, dt, Ls, us)
        rounds.append((Ls, p0s[[3, 12]].copy(), us, np.array(acc), np.array(out5)))
    fs = eb.batch_fused_stats()
    assert fs["members"] == 23 and fs["ranges"] == 11 and fs["launches"] > 0 and fs["timeouts"] == 0, fs
    worst = 0.0
    for q, c in enumerate((3, 12)):
        es.chain_init(x0s[c], low, high)
        for (Ls, p2, us, acc, out5) in rounds:
            a1, o1 = es.chain_trajectory(p2[q], dt, int(Ls[c]), float(us[c]))
            assert bool(a1) == bool(acc[c]), (c, o1, out5[c])
            worst = max(worst, relmax(out5[c], o1))
        worst = max(worst, relmax(eb.batch_get_x(c), es.chain_get_x()))
    print("C2 full size, batch_team_kernel (23 members x 11 ranges, 16 chains) vs the single-chain sweep: %.2e" % worst)
    assert worst < 1e-10
    for e in (eb, es):
        e.close()


def test_c5_share_full_size_oracle_columns(G, orc):
    """The 96 GB share one of 8 GPUs holds of BASELINE configs[4] (200 x 200 observations, the first 3*10^5
    cells of the 200 x 200 x 60 mesh; teamsweep_kernel territory): eight columns of G against the oracle."""
    mesh = G.mesher.PrismMesh((0, 20000.0, 0, 20000.0, 0, 6000.0), (100, 100, 100))
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 20000.0, 200), np.linspace(0, 20000.0, 200))]
    zp = np.zeros_like(xp)
    bounds = mesh.cell_bounds()[: mesh.size // 8]
    N, M = xp.size, bounds.shape[0]
    eng = G.Engine(N, M)
    eng.set_obs(xp, yp, zp)
    eng.set_cells(bounds, 0)
    eng.build_G()
    cols = np.r_[0, 1, 2, M // 2, M // 2 + 1, M - 3, M - 2, M - 1]
    _prism_columns_against_oracle(eng, orc, xp, yp, zp, bounds, cols, "C5 share (4*10^4 x 3*10^5)")
    eng.close()


@pytest.mark.parametrize("form", ["default", "streamed"])
def test_shift_invariant_batch_of_chains_against_the_oracle(G, orc, form, monkeypatch):
    """(streamed: the same with the harmonic store as streaming passes, csrc/lonsymw.hip.h -- every chain's light context
    with its own R^, X^ and D^ partials, on its own stream and thread.)
    BASELINE configs[3] names 8 chains: gh_batch_* on the shift-invariant store runs every chain as a light
    context of its own on the shared tables -- on the harmonic store the chains take turns in the persistent launch
    of csrc/lonres.hip.h, on the direct form each runs on its own stream and thread (tests/test_gpu_mfbatch.py covers
    those).  Coarse geometry: every chain against
    oracle.Problem.leapfrog on the oracle's dense kernel, rounds (gh_batch_trajectory) and lists (gh_batch_run,
    with and without carry-over); C4 geometry: chains 0 and 5 of 8 against a single chain on the table."""
    rng = np.random.default_rng(17)
    if form == "streamed":
        monkeypatch.setenv("GRAVHMC_LONSYM_WIDE", "2")
    mesh, lon, lat, h = _global_model(G, 10.0, 15.0, -1000000, 30000.0)
    N, M = lon.size, mesh.size
    bounds = mesh.cell_bounds()
    K = orc.tess_gz_kernel(lon, lat, h, bounds)
    Aw, wm = orc.col_weight(K)
    d_true = K @ rng.uniform(0.0, 0.5, M)
    dobs = d_true + 0.02 * np.abs(d_true).max() * rng.normal(size=N)
    P = orc.Problem(Aw, dobs, 0.001 * wm, "MS", 0.05, 0.01, wm=wm, shape=mesh.shape)

    def make(lon, lat, h, bounds, shape, dobs, reg, wm_ref):
        t = G.Engine(lon.size, bounds.shape[0])
        t.set_shift_invariant(True)
        t.set_obs(lon, lat, h)
        t.set_cells(bounds, 1, 1.6)
        t.build_G()
        w = t.weight(0.5)
        t.set_data(dobs)
        t.set_reg(reg, 0.05, 0.01, shape, 0.001 * (wm_ref if wm_ref is not None else w))
        return t, w

    t, w = make(lon, lat, h, bounds, mesh.shape, dobs, "MS", wm)
    assert relmax(w, wm) < 1e-11
    assert t.shift_invariant_harmonic()["form"] == ("streamed" if form == "streamed" else "registers")
    C, dt = 4, 0.005
    low, high = 0.0 * wm, 0.8 * wm
    x0s = np.stack([(0.001 + 0.05 * c) * wm for c in range(C)])
    t.batch_init(x0s, low, high)
    xo = [x.copy() for x in x0s]
    worst = 0.0
    for it in range(2):
        Ls = rng.integers(1, 6, size=C)
        p0s = rng.normal(size=(C, M)) * 0.002
        us = rng.uniform(size=C)
        acc, out5 = t.batch_trajectory(p0s, dt, Ls, us)
        for c in range(C):
            xo[c], ao, oo, _ = P.leapfrog(xo[c], p0s[c], dt, int(Ls[c]), low, high, float(us[c]))
            assert bool(acc[c]) == ao
            worst = max(worst, relmax(out5[c], oo), relmax(t.batch_get_x(c), xo[c]))
    for carry in (False, True):
        T = 3
        Ls = rng.integers(1, 6, size=(C, T))
        p0s = rng.normal(size=(C, T, M)) * 0.002
        us = rng.uniform(size=(C, T)) * 0.3
        res = t.batch_run(p0s, dt, Ls, us, want_x=True, carry=carry)
        acc, out5, xs = res[:3]
        if carry:
            assert list(res[3]) == [T] * C and list(res[4]) == [T] * C
        for c in range(C):
            for q in range(T):
                xo[c], ao, oo, _ = P.leapfrog(xo[c], p0s[c, q], dt, int(Ls[c, q]), low, high, float(us[c, q]))
                assert bool(acc[c, q]) == ao
                worst = max(worst, relmax(out5[c, q], oo))
                if ao:
                    worst = max(worst, relmax(xs[c, q], xo[c]))
    print("shift-invariant store [%s], %d chains [coarse] vs ORACLE trajectories: worst %.2e" % (form, C, worst))
    assert worst < 1e-10
    t.close()
    # C4 geometry, 8 chains
    mesh, lon, lat, h = _global_model(G, 3.0, 3.0, -300000, 5000.0)
    M = mesh.size
    rho = np.zeros(mesh.shape)
    rho[1:4, 20:30, 40:60] = 0.3
    tb, wmb = make(lon, lat, h, mesh.cell_bounds(), mesh.shape, np.zeros(lon.size), "Damping", None)
    d = tb.forward(wmb * rho.ravel())
    dobs = d + 0.02 * np.abs(d).max() * rng.normal(size=lon.size)
    tb.set_data(dobs)
    ts, _ = make(lon, lat, h, mesh.cell_bounds(), mesh.shape, dobs, "Damping", None)
    C = 8
    low, high = 0.0 * wmb, 0.8 * wmb
    x0s = np.stack([(0.001 + 0.01 * c) * wmb for c in range(C)])
    tb.batch_init(x0s, low, high)
    T = 2
    Ls = rng.integers(2, 6, size=(C, T))
    p0s = rng.normal(size=(C, T, M)) * 0.001
    us = rng.uniform(size=(C, T))
    acc, out5, _ = tb.batch_run(p0s, 0.005, Ls, us)
    worst = 0.0
    for c in (0, 5):
        ts.chain_init(x0s[c], low, high)
        for q in range(T):
            a1, o1 = ts.chain_trajectory(p0s[c, q], 0.005, int(Ls[c, q]), float(us[c, q]))
            assert bool(a1) == bool(acc[c, q])
            worst = max(worst, relmax(out5[c, q], o1))
        worst = max(worst, relmax(tb.batch_get_x(c), ts.chain_get_x()))
    print("shift-invariant store, 8 chains at C4 size vs the single chain on the table: worst %.2e" % worst)
    assert worst < 1e-10
    tb.close()
    ts.close()


def _run_chain_lists(t, Ls, p0s, us, dt):
    """Engine.run_chain over the given trajectories in ONE library call; per trajectory (accepted, out5, x)."""
    got = []
    t.run_chain(iter([(int(L), p, float(u)) for L, p, u in zip(Ls, p0s, us)]), dt,
                lambda L, acc, o5, x: got.append((bool(acc), np.array(o5), None if x is None else x.copy())),
                want_x=True, batch=len(Ls))
    return got


@pytest.mark.parametrize("case", ["coarse_odd_sizes", "c4_full_size", "two_rows_per_workgroup", "four_rows_per_workgroup"])
def test_shift_invariant_persistent_pass_against_the_oracle(G, orc, case, monkeypatch):
    """lonsymh_resident_kernel (csrc/lonres.hip.h): a batch of trajectories of the chain on the shift-invariant store
    in ONE persistent launch -- the table in the workgroups' registers, forward partials / R^ / Metropolis sums
    exchanged through memory.  coarse (36 longitudes, duplicated +-180 observations, two observation heights,
    shuffled order): every trajectory against oracle.Problem.leapfrog on the oracle's dense tesseroid kernel
    (inversion/hmc.py:85-177 over potential.py:688-810), all four regularisers -- the stencil kinds read their
    neighbours' cells from the positions every workgroup publishes with an evaluation --, decisions included.  C4 at full size
    (BASELINE configs[3]: one chain per GPU): against the same chain on the launches per phase
    (GRAVHMC_LONSYM_RESIDENT=0), which test_shift_invariant_store_against_the_oracle pins to the oracle's rows.  The
    other two: the 3 degree grid with 5 and 15 layers (300 / 900 cell rows: two / four rows per workgroup, the other
    instantiations of the kernel), Damping, against the launches per phase."""
    rng = np.random.default_rng(17)
    if case == "coarse_odd_sizes":
        mesh, lon, lat, h = _global_model(G, 10.0, 15.0, -1000000, 30000.0)
        h[::3] = 45000.0
        perm = rng.permutation(lon.size)
        lon, lat, h = lon[perm], lat[perm], h[perm]
    else:
        dr = {"c4_full_size": -300000, "two_rows_per_workgroup": -600000, "four_rows_per_workgroup": -200000}[case]
        mesh, lon, lat, h = _global_model(G, 3.0, 3.0, dr, 5000.0)
    N, M = lon.size, mesh.size
    bounds = mesh.cell_bounds()

    def engine():
        t = G.Engine(N, M)
        t.set_shift_invariant(True)
        t.set_obs(lon, lat, h)
        t.set_cells(bounds, 1, 1.6)
        t.build_G()
        return t

    t = engine()
    wm = t.weight(0.5)
    rho = np.zeros(mesh.shape)
    rho[:, mesh.shape[1] // 3: mesh.shape[1] // 2, 2: mesh.shape[2] // 3] = 0.4
    Aw = None
    if case == "coarse_odd_sizes":
        K = orc.tess_gz_kernel(lon, lat, h, bounds)
        Aw, wmo = orc.col_weight(K)
        assert relmax(wm, wmo) < 1e-11
        d_true = K @ rho.ravel()
    else:
        t0 = engine()
        d_true = t0.forward(rho.ravel())
        t0.close()
    dobs = d_true + 0.02 * np.abs(d_true).max() * rng.normal(size=N)
    low, high = 0.0 * wm, 0.8 * wm
    nK = 7
    Ls = [int(v) for v in rng.integers(1, 7, nK)]
    Ls[2] = 1                                                  # (a trajectory of a single step)
    p0s = [rng.normal(size=M) * 0.001 for _ in range(nK)]
    us = [float(v) for v in rng.uniform(size=nK)]
    us[4] = 1.0 - 1e-12                                        # (as good as certainly rejected unless H drops)
    for reg in (("Damping",) if case.endswith("per_workgroup") else ("Damping", "MS", "TV", "Smoothness")):
        t.set_data(dobs)
        t.set_reg(reg, 0.05, 0.01, mesh.shape, 0.001 * wm)
        t.chain_init(0.3 * wm * rng.uniform(0.1, 1.0, M), low, high)
        x0 = t.chain_get_x()
        got = _run_chain_lists(t, Ls, p0s, us, 0.005)
        st = t.shift_invariant_resident_stats()
        assert st["workgroups"] > 0 and st["launches"] >= 1 and st["timeouts"] == 0, st
        assert st["evaluations"] >= sum(Ls) + 1
        if case.endswith("per_workgroup"):
            rows = t.shift_invariant_info()["n_rows"]
            assert -(-rows // st["workgroups"]) == (2 if case.startswith("two") else 4), (rows, st)
        x_end = t.chain_get_x()
        if case == "coarse_odd_sizes":
            P = orc.Problem(Aw, dobs, 0.001 * wm, reg, 0.05, 0.01, wm=wm, shape=mesh.shape)
            xo, worst, nacc = x0, 0.0, 0
            for k in range(nK):
                xo, ao, oo, _ = P.leapfrog(xo, p0s[k], 0.005, Ls[k], low, high, us[k])
                assert got[k][0] == ao, (reg, k)
                worst = max(worst, relmax(got[k][1], oo))
                if ao:
                    nacc += 1
                    worst = max(worst, relmax(got[k][2], xo))
            worst = max(worst, relmax(x_end, xo))
            print("persistent harmonic pass [coarse, %s] vs oracle.Problem trajectories: %.2e, %d of %d accepted"
                  % (reg, worst, nacc, nK))
            assert worst < 1e-10
        else:
            monkeypatch.setenv("GRAVHMC_LONSYM_RESIDENT", "0")
            t2 = engine()
            t2.weight(0.5)
            t2.set_data(dobs)
            t2.set_reg(reg, 0.05, 0.01, mesh.shape, 0.001 * wm)
            t2.chain_init(x0, low, high)
            if reg != "Damping":
                # (MS at this step size is stiff in some cells: rounding differences grow ~30x per trajectory -- measured
                # 1.6e-16, 2.3e-13, 1.2e-12, 8.8e-10 ... along the list above -- so every trajectory starts from the
                # reference chain's sample again; Damping runs the whole list in one launch.  The switch is read when a
                # context plans its first run: t has planned, t2 plans under it)
                got, ref, xk = [], [], x0
                for k in range(nK):
                    t.chain_init(xk, low, high)
                    t2.chain_init(xk, low, high)
                    got += _run_chain_lists(t, Ls[k:k + 1], p0s[k:k + 1], us[k:k + 1], 0.005)
                    ref += _run_chain_lists(t2, Ls[k:k + 1], p0s[k:k + 1], us[k:k + 1], 0.005)
                    xk = t2.chain_get_x()
                x_end = t.chain_get_x()
                assert t.shift_invariant_resident_stats()["launches"] >= 1 + nK
            else:
                ref = _run_chain_lists(t2, Ls, p0s, us, 0.005)
            monkeypatch.delenv("GRAVHMC_LONSYM_RESIDENT")
            assert t2.shift_invariant_resident_stats()["launches"] == 0
            worst = 0.0
            for k in range(nK):
                assert got[k][0] == ref[k][0], (reg, k)
                worst = max(worst, relmax(got[k][1], ref[k][1]))
                if ref[k][0]:
                    worst = max(worst, relmax(got[k][2], ref[k][2]))
            worst = max(worst, relmax(x_end, t2.chain_get_x()))
            print("persistent harmonic pass [%s, %s] vs the launches per phase: %.2e, %d of %d accepted"
                  % (case, reg, worst, sum(g[0] for g in got), nK))
            assert worst < 1e-10
            t2.close()
    t.close()


def test_shift_invariant_persistent_pass_gives_up_cleanly(G, orc, monkeypatch):
    """A launch of lonsymh_resident_kernel whose workgroups wait for partners that never come (test hook: eight
    phantom workgroups) times out within its bound (2 s), leaves the chain untouched, and the same call runs on the
    launches per phase: every trajectory still matches oracle.Problem.leapfrog; the next call uses the persistent
    launch again."""
    rng = np.random.default_rng(23)
    mesh, lon, lat, h = _global_model(G, 10.0, 15.0, -1000000, 30000.0)
    N, M = lon.size, mesh.size
    bounds = mesh.cell_bounds()
    t = G.Engine(N, M)
    t.set_shift_invariant(True)
    t.set_obs(lon, lat, h)
    t.set_cells(bounds, 1, 1.6)
    t.build_G()
    wm = t.weight(0.5)
    K = orc.tess_gz_kernel(lon, lat, h, bounds)
    Aw, _ = orc.col_weight(K)
    d_true = K @ rng.uniform(0.0, 0.4, M)
    dobs = d_true + 0.02 * np.abs(d_true).max() * rng.normal(size=N)
    low, high = 0.0 * wm, 0.8 * wm
    t.set_data(dobs)
    t.set_reg("Damping", 0.05, 0.01, mesh.shape, 0.001 * wm)
    t.chain_init(0.2 * wm, low, high)
    P = orc.Problem(Aw, dobs, 0.001 * wm, "Damping", 0.05, 0.01, wm=wm, shape=mesh.shape)
    xo = t.chain_get_x()
    for call in range(3):
        if call == 1:
            monkeypatch.setenv("GRAVHMC_LONRES_TEST_ABORT", "1")
        Ls = [int(v) for v in rng.integers(1, 6, 4)]
        p0s = [rng.normal(size=M) * 0.001 for _ in range(4)]
        us = [float(v) for v in rng.uniform(size=4)]
        got = _run_chain_lists(t, Ls, p0s, us, 0.005)
        monkeypatch.delenv("GRAVHMC_LONRES_TEST_ABORT", raising=False)
        for k in range(4):
            xo, ao, oo, _ = P.leapfrog(xo, p0s[k], 0.005, Ls[k], low, high, us[k])
            assert got[k][0] == ao and relmax(got[k][1], oo) < 1e-10, (call, k)
        assert relmax(t.chain_get_x(), xo) < 1e-10
        st = t.shift_invariant_resident_stats()
        assert st["timeouts"] == (0 if call == 0 else 1) and st["launches"] == (1 if call < 2 else 2), (call, st)
    t.close()


def test_one_degree_global_grid_on_the_streamed_harmonic_store(G, orc):
    """A 1-degree global grid (example/global/SetPMTS.txt's geometry family at 1 degree: 360 x 180 x 10 = 648 000 cells,
    361 x 181 = 65 341 observations; refused by every form of the store until round 4): the dense kernel would be 339 GB,
    the store is T^ = 0.94 GB read twice per leapfrog step (lonsymw.hip.h).  Nothing dense exists to compare with at
    this size: 6 ORACLE rows x 648 000 cells (two polar, four random observations) against K[i, :] = wm * (Aw^T e_i),
    4 oracle columns x 65 341 observations against forward(e_j) and the weights, <A x, r> = <x, A^T r>, linearity of the
    forward, and a chain whose potential the host re-evaluates from forward() at the final state."""
    rng = np.random.default_rng(31)
    mesh, lon, lat, h = _global_model(G, 1.0, 1.0, -300000, 5000.0)
    N, M = lon.size, mesh.size
    assert (N, M) == (65341, 648000)
    bounds = mesh.cell_bounds()
    t = G.Engine(N, M)
    t.set_shift_invariant(True)
    t.set_obs(lon, lat, h)
    t.set_cells(bounds, 1, 1.6)
    t.build_G()
    info, hinfo = t.shift_invariant_info(), t.shift_invariant_harmonic()
    assert info["n_lon"] == 360 and info["n_classes"] == 181 and info["n_rows"] == 1800
    assert hinfo["form"] == "streamed" and hinfo["n_freq"] == 181 and hinfo["table_bytes"] == 900 * 181 * 184 * 16   # (rows of T^ padded to 128-byte lines; ONE row per north-south mirrored pair of cell rows)
    cols = np.r_[0, 359, 324000 + 180, M - 1]
    Kc = orc.tess_gz_kernel(lon, lat, h, bounds[cols])
    e_col = 0.0
    for q, j in enumerate(cols):
        ej = np.zeros(M)
        ej[j] = 1.0
        e_col = max(e_col, float(np.abs(t.forward(ej) - Kc[:, q]).max() / np.abs(Kc[:, q]).max()))
    wt = t.weight(0.5)
    e_w = relmax(wt[cols], np.sqrt((Kc ** 2).sum(0)))
    rows = np.r_[0, 180, rng.choice(N, 4, replace=False)]
    Ko = orc.tess_gz_kernel(lon[rows], lat[rows], h[rows], bounds)
    e_row = 0.0
    for q, i in enumerate(rows):
        ei = np.zeros(N)
        ei[i] = 1.0
        Ki = t.adjoint(ei) * wt
        e_row = max(e_row, float(np.abs(Ki - Ko[q]).max() / np.abs(Ko[q]).max()))
    x1, x2, r = rng.uniform(0, 0.5, M) * wt, rng.uniform(0, 0.5, M) * wt, rng.normal(size=N)
    f1, f2 = t.forward(x1), t.forward(x2)
    e_lin = relmax(t.forward(x1 + 2.0 * x2), f1 + 2.0 * f2)
    e_adj = abs(f1 @ r - x1 @ t.adjoint(r)) / (np.linalg.norm(f1) * np.linalg.norm(r))
    print("1-degree global grid on the streamed harmonic store (%r, %r): 6 ORACLE rows x 648000 cells %.2e, 4 oracle columns %.2e, "
          "weights %.2e, linearity %.2e, adjointness %.2e" % (info, hinfo, e_row, e_col, e_w, e_lin, e_adj))
    assert e_row < 1e-10 and e_col < 1e-10 and e_w < 1e-11 and e_lin < 1e-12 and e_adj < 1e-13
    # a chain: the potential of the final state against a host evaluation from forward()
    rho = np.zeros(mesh.shape)
    rho[2:5, 60:90, 100:160] = 0.3
    dtrue = t.forward(rho.ravel() * wt)
    dobs = dtrue + 0.02 * np.abs(dtrue).max() * rng.normal(size=N)
    t.set_data(dobs)
    t.set_reg("Damping", 0.05, 0.01, mesh.shape, 0.001 * wt)
    t.chain_init(0.001 * wt, 0.0 * wt, 0.8 * wt)
    n_acc = 0
    import time
    t0 = time.perf_counter()
    steps = 0
    for _ in range(4):
        L = int(rng.integers(3, 8))
        acc, o = t.chain_trajectory(rng.normal(size=M) * 0.001, 0.002, L, float(rng.uniform()))
        n_acc += int(acc)
        steps += L
    dt_s = time.perf_counter() - t0
    x = t.chain_get_x()
    U = t.misfit_and_grad(x)
    d = t.forward(x)
    res = (d - d.mean()) - (dobs - dobs.mean())
    phi = float(res @ res)
    assert abs(U[3] - phi) < 1e-10 * phi, (U[3], phi)
    assert relmax(U[1], 2.0 * t.adjoint(res) + 0.05 * 2.0 * (x - 0.001 * wt)) < 1e-10
    print("   chain of 4 trajectories (%d leapfrog steps, %d accepted) in %.3f s incl. the host's momentum uploads" % (steps, n_acc, dt_s))
    assert n_acc > 0
    t.close()
