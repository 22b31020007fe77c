"""Several chains per GPU on the MATRIX-FREE kernel (csrc/mfbatch.hip.h): every entry evaluated by a
pass serves all chains (BASELINE configs[3]: "matrix-free ... 8 chains"; the reference runs its chains
as separate MPI ranks, example/global/run_main.sh:16, inversion/hmc.py:367-369, each re-evaluating
gravmag/_tesseroid_numba.py:32-71 by itself).  Every chain of the batch must reproduce a single-chain
matrix-free context fed the same momenta / lengths / variates, with identical Metropolis decisions."""
import numpy as np
import pytest

from helpers import relmax

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def G(built_lib):
    import gravinv3dhmc_amd as g
    return g


def _tess_problem(G):
    """A coarse global tesseroid model with more rows than one staged chunk (512) and neither N a
    multiple of 64 nor M a multiple of 16; big cells under low observations: a rich near field."""
    mesh = G.mesher.TesseroidMesh((-180, 180, -90, 90, 0, -3000000), (-600000, 10, 10))
    lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, 6.0), np.arange(-87, 88, 6.0), indexing="ij")]
    h = np.full_like(lon, 20000.0)
    b = mesh.cell_bounds()[:-7]             # M = 3233: the last column tile is partial
    return (lon, lat, h), b, 1, (1, 1, b.shape[0])


def _prism_problem(G):
    rng = np.random.default_rng(3)
    N = 1100
    xp, yp = rng.uniform(0, 2000, N), rng.uniform(0, 3000, N)
    mesh = G.mesher.PrismMesh((0, 2000, 0, 3000, 0, 1000), (100, 250, 250))
    b = mesh.cell_bounds()[:-5]
    return (xp, yp, np.zeros(N)), b, 0, (1, 1, b.shape[0])


CASES = {
    # name: (problem, GRAVHMC_MF_EXACT, GRAVHMC_MF_NEAR, regulariser)
    "tess_fast_leaf_near_table": (_tess_problem, "0", "1", "Damping"),
    "tess_reference_order_near_table": (_tess_problem, "1", "1", "MS"),
    "tess_subdivision_inside_the_pass": (_tess_problem, "0", "0", "Damping"),
    "prisms": (_prism_problem, "0", "1", "MS"),
}


@pytest.mark.parametrize("form", ["teams_one_evaluation", "two_passes", "teams_time_out", "teams_time_out_desync"])
@pytest.mark.parametrize("case", list(CASES))
def test_matrix_free_batch_matches_single_chain_matrix_free(G, monkeypatch, case, form):
    """form: the fused team pass (one evaluation per entry and step), the two-pass kernels
    (GRAVHMC_MFB_FUSED=0), or a team pass that gives up (test hook: its members wait for a part that never
    comes) -- the round is then repeated with the two-pass kernels: same chains.  The two time-out paths
    (lock-step round, desynchronised run) are exercised for every one of the four kernel kinds (2 s each)."""
    problem, exact, near, reg = CASES[case]
    monkeypatch.setenv("GRAVHMC_MF_EXACT", exact)
    monkeypatch.setenv("GRAVHMC_MF_NEAR", near)
    monkeypatch.setenv("GRAVHMC_MFB_FUSED", "0" if form == "two_passes" else "1")
    obs, bounds, kind, shape = problem(G)
    N, M = obs[0].size, bounds.shape[0]
    rng = np.random.default_rng(5)
    rho = rng.uniform(0.0, 0.5, M)

    def make(matrix_free=True):
        e = G.Engine(N, M)
        e.set_matrix_free(matrix_free)
        e.set_obs(*obs)
        e.set_cells(bounds, kind, 1.6)
        e.build_G()
        return e

    eb = make()
    d_true = eb.forward(rho)
    wm = eb.weight(0.5)
    dobs = d_true + 0.02 * np.abs(d_true).max() * rng.normal(size=N)

    def finish(e, weighted=True):
        if not weighted:
            e.weight(0.5)
        e.set_data(dobs)
        e.set_reg(reg, 0.05, 0.01, shape, 0.001 * wm)
        return e

    finish(eb)
    st = eb.matrix_free_stats()
    if kind == 1:
        assert (st["near_entries"] > 0) == (near == "1"), st
    C, dt, sig = 5, 0.005, 0.002
    low, high = 0.0 * wm, 0.8 * wm
    x0s = np.stack([(0.001 + 0.05 * c) * wm for c in range(C)])
    eb.batch_init(x0s, low, high)
    # the dense engine's potential at the chains' starting points: the batch's own evaluation
    ed = finish(make(False), weighted=False)
    singles = []
    for c in range(C):
        e = finish(make(), weighted=False)
        e.chain_init(x0s[c], low, high)
        singles.append(e)
    n_acc = n_rej = 0
    worst = 0.0
    fs0 = eb.batch_fused_stats()
    assert (fs0["members"] > 0) == (form != "two_passes"), fs0
    for it in range(3):
        Ls = rng.integers(1, 6, size=C)
        p0s = rng.normal(size=(C, M)) * sig
        us = rng.uniform(size=C) * (0.05 if it == 1 else 1.0)
        if form == "teams_time_out" and it == 1:
            monkeypatch.setenv("GRAVHMC_MFB_TEST_ABORT", "1")
        acc, out5 = eb.batch_trajectory(p0s, dt, Ls, us)
        monkeypatch.delenv("GRAVHMC_MFB_TEST_ABORT", raising=False)
        for c in range(C):
            a1, o1 = singles[c].chain_trajectory(p0s[c], dt, int(Ls[c]), float(us[c]))
            assert a1 == acc[c], (case, it, c, o1, out5[c])
            worst = max(worst, relmax(out5[c], o1), relmax(eb.batch_get_x(c), singles[c].chain_get_x()))
            n_acc += a1
            n_rej += not a1
    fs = eb.batch_fused_stats()
    print("matrix-free batch [%s, %s]: N %d M %d, %d chains, accepted %d rejected %d, worst deviation from the "
          "single-chain matrix-free engine %.2e; near field %r; team form %r" % (case, form, N, M, C, n_acc, n_rej, worst, st, fs))
    assert worst < 1e-10 and n_acc > 0
    if form == "teams_one_evaluation":
        assert fs["launches"] > 0 and fs["timeouts"] == 0
    if form == "teams_time_out":
        assert fs["timeouts"] == 1 and fs["members"] == 0     # gave up once, two-pass kernels from there on
    if form == "teams_time_out_desync":
        # a team pass gives up in the middle of gh_batch_run: the trajectories in flight are replayed from
        # their chains' current states and their own momenta (kept on the device), on the two-pass kernels
        monkeypatch.setenv("GRAVHMC_MFB_TEST_ABORT", "1")
    # against the DENSE single-chain engine as well (stored G: the reference's formulation)
    a = ed.misfit_and_grad(eb.batch_get_x(2))
    b = singles[2].misfit_and_grad(singles[2].chain_get_x())
    assert abs(a[0] - b[0]) < 1e-10 * abs(b[0]) and relmax(a[1], b[1]) < 1e-10
    # desynchronised scheduling (gh_batch_run: a chain starts its next trajectory in the sweep after it
    # finished the previous one; speculative first steps) computes what lock-step rounds compute
    T = 3
    Ls = rng.integers(1, 6, size=(C, T))
    p0s = rng.normal(size=(C, T, M)) * sig
    us = rng.uniform(size=(C, T))
    acc_a, out_a, xs_a = eb.batch_run(p0s, dt, Ls, us, want_x=True)
    monkeypatch.delenv("GRAVHMC_MFB_TEST_ABORT", raising=False)
    if form == "teams_time_out_desync":
        fs = eb.batch_fused_stats()
        assert fs["timeouts"] == 1 and fs["members"] == 0, fs
    for t in range(T):
        for c in range(C):
            a1, o1 = singles[c].chain_trajectory(p0s[c, t], dt, int(Ls[c, t]), float(us[c, t]))
            assert bool(a1) == bool(acc_a[c, t]), (case, c, t)
            assert relmax(out_a[c, t], o1) < 1e-10
            if a1:
                assert relmax(xs_a[c, t], singles[c].chain_get_x()) < 1e-10
    for c in range(C):
        assert relmax(eb.batch_get_x(c), singles[c].chain_get_x()) < 1e-10
    for e in singles + [eb, ed]:
        e.close()


def test_c4_matrix_free_eight_chains_full_size(G):
    """BASELINE configs[3] at full size on one GPU: the 7381 x 72000 global tesseroid model, matrix-free,
    EIGHT chains sharing every evaluated entry, against the single-chain matrix-free engine (chains 0, 3
    and 7 re-run one by one) and -- potential and gradient at the final states -- the dense engine."""
    mesh = G.mesher.TesseroidMesh((-180, 180, -90, 90, 0, -3000000), (-300000, 3, 3))
    lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, 3.0), np.arange(-90, 91, 3.0), indexing="ij")]
    h = np.full_like(lon, 5000.0)
    N, M = lon.size, mesh.size
    rho = np.zeros(mesh.shape)
    rho[1:4, 20:30, 40:60] = 0.3
    rho = rho.ravel()
    rng = np.random.default_rng(45)

    def make(matrix_free):
        e = G.Engine(N, M)
        e.set_matrix_free(matrix_free)
        e.set_obs(lon, lat, h)
        e.set_cells(mesh.cell_bounds(), 1, 1.6)
        e.build_G()
        return e

    eb, es, ed = make(True), make(True), make(False)
    dtrue = ed.forward(rho)
    wm = ed.weight(0.5)
    assert relmax(eb.weight(0.5), wm) < 1e-12 and relmax(es.weight(0.5), wm) < 1e-12
    dobs = dtrue + 0.02 * np.abs(dtrue).max() * rng.normal(size=N)
    for e in (eb, es, ed):
        e.set_data(dobs)
        e.set_reg("Damping", 0.05, 0.01, mesh.shape, 0.001 * wm)
    C, dt = 8, 0.005
    low, high = 0.0 * wm, 0.8 * wm
    x0s = np.stack([(0.001 + 0.01 * c) * wm for c in range(C)])
    eb.batch_init(x0s, low, high)
    rounds = []
    for it in range(2):
        Ls = rng.integers(2, 6, size=C)
        p0s = rng.normal(size=(C, M)) * 0.001
        us = rng.uniform(size=C)
        acc, out5 = eb.batch_trajectory(p0s, dt, Ls, us)
        rounds.append((Ls, p0s, us, np.array(acc), np.array(out5)))
    worst = 0.0
    for c in (0, 3, 7):
        es.chain_init(x0s[c], low, high)
        for (Ls, p0s, us, acc, out5) in rounds:
            a1, o1 = es.chain_trajectory(p0s[c], dt, int(Ls[c]), float(us[c]))
            assert a1 == acc[c], (c, o1, out5[c])
            worst = max(worst, relmax(out5[c], o1))
        xb = eb.batch_get_x(c)
        worst = max(worst, relmax(xb, es.chain_get_x()))
        a, b = es.misfit_and_grad(xb), ed.misfit_and_grad(xb)
        assert abs(a[0] - b[0]) < 1e-10 * abs(b[0]) and relmax(a[1], b[1]) < 1e-10
    print("C4 matrix-free, 8 chains per GPU vs the single-chain matrix-free engine: worst %.2e, accepted %d of %d"
          % (worst, sum(int(np.sum(r[3])) for r in rounds), 2 * C))
    assert worst < 1e-10
    for e in (eb, es, ed):
        e.close()


# ------------------------------------------------------------------ shift-invariant store (lonsym.hip.h)

def _global_model(G, dlon, dlat, dr, obs_h, nlat_obs_step=None):
    mesh = G.mesher.TesseroidMesh((-180, 180, -90, 90, 0, -3000000), (dr, dlat, dlon))
    lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, dlon), np.arange(-90, 91, nlat_obs_step or dlat),
                                               indexing="ij")]
    return mesh, lon, lat, np.full_like(lon, obs_h)


@pytest.mark.parametrize("case", ["coarse_odd_sizes", "c4_full_size", "c4_full_size_direct_correlations",
                                  "c4_full_size_one_launch_epilogue", "coarse_odd_sizes_streamed", "c4_full_size_streamed",
                                  "lon180_streamed", "classes91_streamed", "odd45", "odd45_streamed",
                                  "selfmirror_streamed", "c4_full_size_streamed_no_mirror"])
def test_shift_invariant_store_matches_the_stored_kernel(G, monkeypatch, case):
    """gh_set_shift_invariant: K[i, (c, k)] = T[c][class_i][(m_i - k) mod n] for regular spherical grids
    (example/global/main_global.py:25-28; BASELINE configs[3]'s geometry) against the dense engine on the
    same problem: column norms, unweighted forward (the reference's gz), adjoint, potential + gradient
    for a cell-local and a stencil regulariser, and a chain with identical decisions.  coarse: 36 longitudes
    (not a multiple of 8 per block boundary: 36 = 4.5 blocks), duplicated +-180 observations, shuffled
    observation order, two observation heights (classes = (lat, h) pairs).  *_streamed: the harmonic form as
    streaming passes over T^ (lonsymw.hip.h) -- forced on the two geometries above, and chosen by the library
    where the register form does not apply: 180 longitudes (91 frequencies > 64) and 91 observation classes (> 64)."""
    rng = np.random.default_rng(12)
    # (default: the longitude-harmonic form with its three-launch epilogue; the direct correlations of round 3 and the
    # one-launch epilogue -- built, not the default, DESIGN 4.9 -- stay covered)
    if case.endswith("direct_correlations"):
        monkeypatch.setenv("GRAVHMC_LONSYM_HARMONIC", "0")
    if case.endswith("one_launch_epilogue"):
        monkeypatch.setenv("GRAVHMC_LONSYM_FUSED", "1")
    if case in ("coarse_odd_sizes_streamed", "c4_full_size_streamed", "odd45_streamed", "selfmirror_streamed",
                "c4_full_size_streamed_no_mirror"):
        monkeypatch.setenv("GRAVHMC_LONSYM_WIDE", "2")
    if case.endswith("_no_mirror"):      # (one row of T^ per cell row although the grid is symmetric about the equator)
        monkeypatch.setenv("GRAVHMC_LW_MIRROR", "0")
    if case.startswith("coarse_odd_sizes"):
        mesh, lon, lat, h = _global_model(G, 10.0, 15.0, -1000000, 30000.0)
        h[::3] = 45000.0
        perm = rng.permutation(lon.size)
        lon, lat, h = lon[perm], lat[perm], h[perm]
        tol = 1e-10
    elif case == "selfmirror_streamed":
        # nine latitude bands of 20 degrees: the band on the equator is its own north-south mirror image
        mesh, lon, lat, h = _global_model(G, 10.0, 20.0, -1000000, 30000.0)
        tol = 1e-10
    elif case.startswith("odd45"):
        mesh, lon, lat, h = _global_model(G, 8.0, 15.0, -1000000, 30000.0)     # an ODD number of longitudes: no pair (n / 2, n / 2)
        tol = 1e-10
    elif case == "lon180_streamed":
        mesh, lon, lat, h = _global_model(G, 2.0, 3.0, -1500000, 5000.0)       # 180 longitudes, 61 classes, 120 cell rows
        tol = 1e-10
    elif case == "classes91_streamed":
        mesh, lon, lat, h = _global_model(G, 4.0, 2.0, -1500000, 5000.0)       # 90 longitudes, 91 classes, 180 cell rows
        tol = 1e-10
    else:
        mesh, lon, lat, h = _global_model(G, 3.0, 3.0, -300000, 5000.0)
        tol = 1e-10
    N, M = lon.size, mesh.size
    rho = rng.uniform(0.0, 0.5, M)
    engs = {}
    for tag in ("dense", "table"):
        e = G.Engine(N, M)
        if tag == "table":
            e.set_shift_invariant(True)
        e.set_obs(lon, lat, h)
        e.set_cells(mesh.cell_bounds(), 1, 1.6)
        e.build_G()
        engs[tag] = e
    d, t = engs["dense"], engs["table"]
    info = t.shift_invariant_info()
    assert info["n_lon"] == mesh.shape[2] and info["n_rows"] == mesh.shape[0] * mesh.shape[1]
    form = t.shift_invariant_harmonic()["form"]
    assert form == ("streamed" if "_streamed" in case else None if case.endswith("direct_correlations") else "registers"), form
    if form == "streamed":
        # the north-south mirror halves the table on grids symmetric about the equator (one row per pair, the equator's alone)
        nrows, full = info["n_rows"], t.shift_invariant_harmonic()["table_bytes"]
        per_row = info["n_classes"] * ((mesh.shape[2] // 2 + 1 + 7) // 8 * 8) * 16
        expect = {"selfmirror_streamed": mesh.shape[0] * 5, "c4_full_size_streamed_no_mirror": nrows}
        assert full == per_row * expect.get(case, nrows // 2), (case, full, per_row, nrows)
    assert info["table_bytes"] < N * M * 8 / 10
    dtrue = d.forward(rho)
    e_fwd = relmax(t.forward(rho), dtrue)
    wd, wt = d.weight(0.5), t.weight(0.5)
    e_w = relmax(wt, wd)
    r = rng.normal(size=N)
    e_adj = relmax(t.adjoint(r), d.adjoint(r))
    dobs = dtrue + 0.02 * np.abs(dtrue).max() * rng.normal(size=N)
    x = rng.uniform(0, 0.8, M) * wd
    worst = 0.0
    for reg in ("Damping", "TV"):
        for e in (d, t):
            e.set_data(dobs)
            e.set_reg(reg, 0.05, 0.01, mesh.shape, 0.001 * wd)
        a, b = t.misfit_and_grad(x), d.misfit_and_grad(x)
        worst = max(worst, abs(a[0] - b[0]) / abs(b[0]), relmax(a[1], b[1]), relmax(a[2], b[2]))
    print("shift-invariant store [%s, harmonic form %s]: %r; vs dense: forward %.2e weights %.2e adjoint %.2e potential/gradient %.2e"
          % (case, form, info, e_fwd, e_w, e_adj, worst))
    assert e_fwd < tol and e_w < tol and e_adj < tol and worst < tol
    trajs = [(int(rng.integers(2, 7)), rng.normal(size=M) * 0.001, float(rng.uniform())) for _ in range(5)]
    outs = {}
    for tag, e in engs.items():
        e.chain_init(0.001 * wd, 0.0 * wd, 0.8 * wd)
        res = []
        e.run_chain(iter(trajs), 0.005, lambda L, acc, o, xx, res=res: res.append((acc, o.copy())))
        outs[tag] = (res, e.chain_get_x())
    for (a1, o1), (a2, o2) in zip(outs["table"][0], outs["dense"][0]):
        assert a1 == a2 and relmax(o1, o2) < 1e-9
    assert relmax(outs["table"][1], outs["dense"][1]) < 1e-9
    for e in engs.values():
        e.close()


def test_shift_invariant_store_refuses_irregular_geometry(G):
    """Geometry without the structure: gh_build_G fails with the reason (no silent fallback)."""
    mesh, lon, lat, h = _global_model(G, 10.0, 15.0, -1000000, 30000.0)
    for what in ("obs_off_grid", "half_circle", "prisms"):
        e = G.Engine(lon.size, mesh.size)
        e.set_shift_invariant(True)
        b = mesh.cell_bounds().copy()
        lo = lon.copy()
        if what == "obs_off_grid":
            lo[5] += 1.234
        if what == "half_circle":
            b[:, 0:2] *= 0.5
        e.set_obs(lo, lat, h)
        e.set_cells(b, 0 if what == "prisms" else 1, 1.6)
        with pytest.raises(NotImplementedError, match="shift-invariant"):
            e.build_G()
        e.close()


def test_hmcsample_batch_on_a_matrix_free_global_model(G, tmp_path, capsys):
    """The reference's `mpiexec -n K python main_global.py` (example/global/run_main.sh:16) on ONE GPU without
    a stored kernel: HMCSampleBatch over a matrix-free spherical GravMagModule -- chains = ranks 0..2, each
    its own legacy RandomState(seed + rank), folder and console lines -- against separate HMCSample(myrank=r)
    runs on the DENSE module: the same console lines (7 printed digits), the same model.dat rows."""
    rng = np.random.default_rng(21)
    mrange, mspacing = (-180, 180, -90, 90, 0, -3000000), (-1000000, 15, 10)
    lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, 10.0), np.arange(-90, 91, 15.0), indexing="ij")]
    h = np.full_like(lon, 30000.0)
    mesh = G.mesher.TesseroidMesh(mrange, mspacing)
    M, N = mesh.size, lon.size
    probe = G.GravMagModule(np.zeros(N), mrange, mspacing, (lon, lat, h), coordinate="spherical", verbose=False)
    rho = np.zeros(mesh.shape)
    rho[1:, 4:8, 10:20] = 0.3
    dobs = probe._engine.forward(probe.Wm.diagonal() * rho.ravel()) * (1.0 + 0.01 * rng.normal(size=N))
    probe._engine.close()
    args = (np.full(M, 0.001), np.full(M, 0.001), np.c_[np.zeros(M), np.full(M, 0.8)], "mandatory", 1000, dobs,
            "Fixed", 0.8, 0.05, "Damping", 0.01, 100, 0.001)
    nsamp, dt = 6, 0.005
    gm = G.GravMagModule(dobs, mrange, mspacing, (lon, lat, h), coordinate="spherical", verbose=False,
                         matrix_free=True)
    capsys.readouterr()
    G.HMCSampleBatch(gm, 3, nsamp, 0, dt, [5, 20], *args, save_folder=str(tmp_path / "mfbatch_chain"))
    out = capsys.readouterr().out.splitlines()
    fs = gm._engine.batch_fused_stats()
    assert fs["launches"] > 0 and fs["timeouts"] == 0, fs
    for r in range(3):
        gs = G.GravMagModule(dobs, mrange, mspacing, (lon, lat, h), coordinate="spherical", verbose=False)
        capsys.readouterr()
        G.HMCSample(gs, nsamp, 0, dt, [5, 20], *args, myrank=r, save_folder=str(tmp_path / "dense_chain"))
        single = [l for l in capsys.readouterr().out.splitlines() if l.startswith("chain %d:" % r)]
        assert [l for l in out if l.startswith("chain %d:" % r)] == single
        a = np.loadtxt(str(tmp_path / ("mfbatch_chain%d" % r)) + "/model.dat")
        b = np.loadtxt(str(tmp_path / ("dense_chain%d" % r)) + "/model.dat")
        assert a.shape == (nsamp, M)
        np.testing.assert_allclose(a, b, atol=2e-8)
        gs._engine.close()
    gm._engine.close()


@pytest.mark.parametrize("wavelet", ["3D", "1D"])
def test_wavelet_forward_on_a_matrix_free_model(G, wavelet):
    """wavelet='1D'/'3D' together with matrix_free=True (refused until round 3): the compressor's rows
    (compressor3D.kernelcompressor transforms whole rows of Aw, gravmag/compressor3D.py:17-44) are evaluated
    instead of gathered from a stored kernel; the forward then runs on the CSR operator, the gradient on the
    matrix-free adjoint (potential.py:693-708: compressed forward, exact dense adjoint).  Against the same
    module with the stored kernel: the same CSR, potential, gradient and chain."""
    from conftest import gold
    p = gold("potential_small.npz")
    mods = {}
    for tag, mf in (("stored", False), ("matrix_free", True)):
        mods[tag] = G.GravMagModule(p["dobs"], tuple(p["mrange"]), tuple(p["mspacing"]), (p["xp"], p["yp"], p["zp"]),
                                    verbose=False, wavelet=wavelet, matrix_free=mf)
    a, b = mods["matrix_free"], mods["stored"]
    ca, cb = a.Awcp, b.Awcp
    assert ca.shape == cb.shape and ca.nnz == cb.nnz and np.array_equal(ca.indices, cb.indices)
    assert relmax(ca.data, cb.data) < 1e-12
    wm = b.Wm.diagonal()
    rng = np.random.default_rng(8)
    for reg in ("MS", "TV"):
        x = rng.uniform(0, 1, wm.size) * wm
        ra = a.misfit_and_grad(x, p["mwapr"], None, None, "mandatory", 1000, 0.7, regulization=reg, beta=0.001)
        rb = b.misfit_and_grad(x, p["mwapr"], None, None, "mandatory", 1000, 0.7, regulization=reg, beta=0.001)
        assert abs(ra[0] - rb[0]) < 1e-12 * abs(rb[0]) and relmax(ra[1], rb[1]) < 1e-11 and relmax(ra[2], rb[2]) < 1e-11
    trajs = [(int(rng.integers(1, 9)), rng.normal(size=wm.size) * 0.3, float(rng.uniform())) for _ in range(6)]
    outs = []
    for m in (a, b):
        e = m._engine
        e.set_reg("MS", 1.0, 0.001, p["shape"], 0.001 * wm)
        e.chain_init(0.001 * wm, 0.0 * wm, 0.02 * wm)
        res = []
        e.run_chain(iter(trajs), 0.02, lambda L, acc, o, x, res=res: res.append((acc, o.copy())))
        outs.append((res, e.chain_get_x()))
    for (a1, o1), (a2, o2) in zip(outs[0][0], outs[1][0]):
        assert a1 == a2 and relmax(o1, o2) < 1e-10
    assert relmax(outs[0][1], outs[1][1]) < 1e-10
    for m in mods.values():
        m._engine.close()


def test_single_chain_matrix_free_on_teams_and_its_time_out(G, monkeypatch):
    """mf_team_kernel: the leapfrog steps of ONE matrix-free tesseroid chain on teams of workgroups (a wave
    keeps its column's dot as one number per tile; 16 doubles per member and tile cross the team) against the
    column-per-workgroup pass (GRAVHMC_MF_TEAM=0) and the dense engine: the same chain; and a team pass that
    gives up (test hook) -- the trajectory is repeated on the column-per-workgroup pass, the teams stay off."""
    obs, bounds, kind, shape = _tess_problem(G)
    N, M = obs[0].size, bounds.shape[0]
    rng = np.random.default_rng(15)
    rho = rng.uniform(0.0, 0.5, M)
    trajs = [(int(rng.integers(1, 7)), rng.normal(size=M) * 0.002, float(rng.uniform())) for _ in range(8)]

    def run(matrix_free, team, abort_at=None):
        monkeypatch.setenv("GRAVHMC_MF_TEAM", team)
        e = G.Engine(N, M)
        e.set_matrix_free(matrix_free)
        e.set_obs(*obs)
        e.set_cells(bounds, kind, 1.6)
        e.build_G()
        d_true = e.forward(rho)
        wm = e.weight(0.5)
        e.set_data(d_true + 0.02 * np.abs(d_true).max() * np.random.default_rng(16).normal(size=N))
        e.set_reg("MS", 0.05, 0.01, shape, 0.001 * wm)
        e.chain_init(0.001 * wm, 0.0 * wm, 0.8 * wm)
        res = []
        for k, (L, p0, u) in enumerate(trajs):
            if abort_at == k:
                monkeypatch.setenv("GRAVHMC_MF_TEAM_TEST_ABORT", "1")
            acc, o = e.chain_trajectory(p0, 0.005, L, u)
            monkeypatch.delenv("GRAVHMC_MF_TEAM_TEST_ABORT", raising=False)
            res.append((acc, np.array(o)))
        st = e.matrix_free_team_stats() if matrix_free else None
        x = e.chain_get_x()
        e.close()
        return res, x, st

    team, xt, st_t = run(True, "1")
    plain, xp_, st_p = run(True, "0")
    dense, xd, _ = run(False, "0")
    tout, xo, st_o = run(True, "1", abort_at=3)
    assert st_t["launches"] > 0 and st_t["timeouts"] == 0 and st_t["members"] > 0, st_t
    assert st_p["launches"] == 0, st_p
    assert st_o["timeouts"] == 1 and st_o["members"] == 0, st_o
    worst = 0.0
    for other, xo_ in ((plain, xp_), (dense, xd), (tout, xo)):
        for (a1, o1), (a2, o2) in zip(team, other):
            assert a1 == a2
            worst = max(worst, relmax(o1, o2))
        worst = max(worst, relmax(xt, xo_))
    print("single-chain matrix-free on teams %r vs column-per-workgroup pass / dense engine / after a time-out: %.2e"
          % (st_t, worst))
    assert worst < 1e-10


@pytest.mark.parametrize("wavelet", ["3D", "1D"])
def test_wavelet_forward_on_the_shift_invariant_store(G, wavelet):
    """wavelet='1D'/'3D' together with shift_invariant=True (refused until round 4): the compressor's rows are
    evaluated by the matrix-free row kernel (the store never holds G), the forward runs on the CSR operator, the
    gradient on the table's adjoint (potential.py:693-708: compressed forward, exact adjoint).  Against the same
    module with the stored kernel: the same CSR, potential, gradient and chain; batched chains are refused."""
    rng = np.random.default_rng(23)
    mrange, mspacing = (-180, 180, -90, 90, 0, -3000000), (-1000000, 15, 10)
    lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, 10.0), np.arange(-90, 91, 15.0), indexing="ij")]
    h = np.full_like(lon, 30000.0)
    mesh = G.mesher.TesseroidMesh(mrange, mspacing)
    M, N = mesh.size, lon.size
    probe = G.GravMagModule(np.zeros(N), mrange, mspacing, (lon, lat, h), coordinate="spherical", verbose=False)
    rho = np.zeros(mesh.shape)
    rho[1:, 4:8, 10:20] = 0.3
    dobs = probe._engine.forward(probe.Wm.diagonal() * rho.ravel()) * (1.0 + 0.01 * rng.normal(size=N))
    probe._engine.close()
    mods = {}
    for tag, si in (("stored", False), ("table", True)):
        mods[tag] = G.GravMagModule(dobs, mrange, mspacing, (lon, lat, h), coordinate="spherical", verbose=False,
                                    wavelet=wavelet, shift_invariant=si)
    a, b = mods["table"], mods["stored"]
    assert a._engine.shift_invariant_info()["n_lon"] == mesh.shape[2]
    ca, cb = a.Awcp, b.Awcp
    assert ca.shape == cb.shape and ca.nnz == cb.nnz and np.array_equal(ca.indices, cb.indices)
    assert relmax(ca.data, cb.data) < 1e-11
    wm = b.Wm.diagonal()
    mwapr = 0.001 * wm
    worst = 0.0
    for reg in ("MS", "TV"):
        x = rng.uniform(0, 0.8, M) * wm
        ra = a.misfit_and_grad(x, mwapr, None, None, "mandatory", 1000, 0.7, regulization=reg, beta=0.001)
        rb = b.misfit_and_grad(x, mwapr, None, None, "mandatory", 1000, 0.7, regulization=reg, beta=0.001)
        worst = max(worst, abs(ra[0] - rb[0]) / abs(rb[0]), relmax(ra[1], rb[1]), relmax(ra[2], rb[2]))
    assert worst < 1e-10, worst
    trajs = [(int(rng.integers(1, 9)), rng.normal(size=M) * 0.001, float(rng.uniform())) for _ in range(6)]
    outs = []
    for m in (a, b):
        e = m._engine
        e.set_reg("MS", 0.05, 0.001, mesh.shape, mwapr)
        e.chain_init(0.001 * wm, 0.0 * wm, 0.8 * wm)
        res = []
        e.run_chain(iter(trajs), 0.005, lambda L, acc, o, x, res=res: res.append((acc, o.copy())))
        outs.append((res, e.chain_get_x()))
    for (a1, o1), (a2, o2) in zip(outs[0][0], outs[1][0]):
        assert a1 == a2 and relmax(o1, o2) < 1e-10
    assert relmax(outs[0][1], outs[1][1]) < 1e-10
    print("wavelet %s forward on the shift-invariant store vs the stored kernel: nnz %d, potential/gradient %.2e, chain of %d "
          "trajectories identical decisions" % (wavelet, ca.nnz, worst, len(trajs)))
    with pytest.raises(NotImplementedError, match="wavelet"):
        a._engine.batch_init(np.stack([0.001 * wm, 0.002 * wm]), 0.0 * wm, 0.8 * wm)
    for m in mods.values():
        m._engine.close()


@pytest.mark.parametrize("form", ["registers", "streamed"])
def test_hmcsample_on_the_shift_invariant_store(G, tmp_path, capsys, monkeypatch, form):
    """The reference's global example as a user runs it (example/global/main_global.py:25-60: GravMagModule(coordinate=
    'spherical') -> HMCSample) with shift_invariant=True -- the sampler's calls reach the store's passes (registers: the
    persistent launch of csrc/lonres.hip.h; streamed: the four launches of csrc/lonsymw.hip.h, forced on this small
    geometry) -- against the same run on the DENSE module: the same console lines (7 printed digits), the same model.dat
    and misfit.dat rows."""
    if form == "streamed":
        monkeypatch.setenv("GRAVHMC_LONSYM_WIDE", "2")
    rng = np.random.default_rng(27)
    mrange, mspacing = (-180, 180, -90, 90, 0, -3000000), (-1000000, 15, 10)
    lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, 10.0), np.arange(-90, 91, 15.0), indexing="ij")]
    h = np.full_like(lon, 30000.0)
    mesh = G.mesher.TesseroidMesh(mrange, mspacing)
    M, N = mesh.size, lon.size
    dense = G.GravMagModule(np.zeros(N), mrange, mspacing, (lon, lat, h), coordinate="spherical", verbose=False)
    rho = np.zeros(mesh.shape)
    rho[1:, 4:8, 10:20] = 0.3
    dobs = dense._engine.forward(dense.Wm.diagonal() * rho.ravel()) * (1.0 + 0.01 * rng.normal(size=N))
    dense._engine.close()
    args = (np.full(M, 0.001), np.full(M, 0.001), np.c_[np.zeros(M), np.full(M, 0.8)], "mandatory", 1000, dobs,
            "Fixed", 0.8, 0.05, "Damping", 0.01, 100, 0.001)
    nsamp, dt = 8, 0.005
    outs = {}
    for tag in ("table", "dense"):
        gm = G.GravMagModule(dobs, mrange, mspacing, (lon, lat, h), coordinate="spherical", verbose=False,
                             shift_invariant=(tag == "table"))
        if tag == "table":
            assert gm._engine.shift_invariant_harmonic()["form"] == form
        capsys.readouterr()
        G.HMCSample(gm, nsamp, 0, dt, [5, 20], *args, myrank=0, save_folder=str(tmp_path / tag))
        lines = [l for l in capsys.readouterr().out.splitlines() if l.startswith("chain 0:")]
        folder = str(tmp_path / tag) + "0"
        outs[tag] = (lines, np.loadtxt(folder + "/model.dat"), np.loadtxt(folder + "/misfit.dat"))
        if tag == "table" and form == "registers":
            st = gm._engine.shift_invariant_resident_stats()
            assert st["launches"] > 0 and st["timeouts"] == 0, st
        gm._engine.close()
    assert len(outs["table"][0]) > 0 and outs["table"][0] == outs["dense"][0]
    assert outs["table"][1].shape == (nsamp, M)
    np.testing.assert_allclose(outs["table"][1], outs["dense"][1], atol=2e-8)
    np.testing.assert_allclose(outs["table"][2], outs["dense"][2], rtol=1e-7)
