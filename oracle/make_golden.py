"""Generate tests/golden/*.npz from the reference itself and pin the oracle against it.

TEST INFRASTRUCTURE ONLY; runs in the build container only (needs /root/reference):
    python -m oracle.make_golden
The fixtures are DATA (inputs + expected outputs + lines of the reference's committed run
logs + the reference's example input data files); no reference source is stored.
Every block first checks the C oracle (oracle/gravhmc_oracle.c) against the reference
output it is about to store and fails loudly on disagreement (tolerances inline).
"""
import io
import os
import re
import sys
import contextlib

import numpy as np

from oracle import oracle, ref_harness

GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests",
                    "golden")
REF = ref_harness.REF


def _quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = fn(*a, **k)
    return out


def _relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300))


def prism_cases(R):
    """Singular and regular (obs, cell) geometries straight through _prism.gz."""
    cells = np.array([[0, 100, 0, 100, 0, 100], [-50, 50, -30, 70, 10, 60],
                      [1000, 1100, 2000, 2100, 900, 1000], [0, 100, 0, 100, 100, 300.5]],
                     dtype=float)
    pts = []
    for x in (-100.0, 0.0, 50.0, 100.0, 250.0):
        for y in (-100.0, 0.0, 50.0, 100.0, 180.0):
            for z in (0.0, -10.0, 100.0, 50.0):
                pts.append((x, y, z))
    pts += [(1e4, 2e4, 0.0), (-3e3, 5.0, -200.0), (1050.0, 2050.0, 0.0), (33.3, 66.6, -0.01)]
    pts = np.array(pts)
    xp, yp, zp = [np.ascontiguousarray(pts[:, i]) for i in range(3)]
    K = np.zeros((len(pts), len(cells)))
    for c, b in enumerate(cells):
        res = np.zeros(len(pts))
        k1 = np.zeros(len(pts))
        R._prism.gz(xp, yp, zp, *[float(v) for v in b], 1.0, res, k1)
        K[:, c] = k1
    K *= R.constants.G * R.constants.SI2MGAL
    Ko = oracle.prism_gz_kernel(xp, yp, zp, cells)
    assert np.array_equal(Ko, K), "oracle prism kernel differs from reference"
    np.savez_compressed(os.path.join(GOLD, "prism_cases.npz"), xp=xp, yp=yp, zp=zp,
                        cells=cells, K=K)
    print("prism_cases", K.shape, "finite:", np.isfinite(K).all())


def c1_spot(R):
    """Config C1 (uniformgrid singlecube) kernel spot values + noise-free forward."""
    mesh = _quiet(R.mesher.PrismMesh, (0, 2000, 0, 3000, 0, 1000), (100, 100, 100))
    nx, ny, nz = 20, 30, 10
    rho = np.zeros(mesh.size)
    for iz in range(nz):
        for iy in range(ny):
            for ix in range(nx):
                if 7 <= ix <= 10 and 10 <= iy <= 17 and 2 <= iz <= 4:
                    rho[nx * ny * iz + nx * iy + ix] = 1.0
    mesh.addprop('density', rho)
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 3000, 30), np.linspace(0, 2000, 20))]
    zp = np.zeros_like(xp)
    gz, K = R.prism.gz(xp, yp, zp, mesh)
    b = np.array([c.get_bounds() for c in mesh])
    Ko = oracle.prism_gz_kernel(xp, yp, zp, b)
    assert np.array_equal(Ko, K)
    rng = np.random.default_rng(7)
    ii = rng.integers(0, 600, 256)
    jj = rng.integers(0, 6000, 256)
    np.savez_compressed(os.path.join(GOLD, "c1_spot.npz"), gz=gz, rho=rho,
                        colnorm=np.sqrt((K ** 2).sum(0)), rownorm=np.sqrt((K ** 2).sum(1)),
                        ii=ii, jj=jj, Kij=K[ii, jj], fro=np.linalg.norm(K),
                        K_col0=K[:, 0], K_col5999=K[:, 5999], K_row0=K[0, :])
    print("c1_spot gz sum", gz.sum(), "fro", np.linalg.norm(K))
    return xp, yp, zp, K


def tess_cases(R):
    """Coarse global mesh + near-field / polar / thin-cell geometries."""
    mesh = _quiet(R.mesher.TesseroidMesh, (-180, 180, -90, 90, 0, -3e6), (-300000, 30, 30))
    rho = np.linspace(0.1, 0.5, mesh.size)
    mesh.addprop('density', rho)
    lon, lat = [a.ravel() for a in np.meshgrid(np.linspace(-180, 177, 6), np.linspace(-90, 90, 5))]
    h = np.full_like(lon, 5000.0)
    gz, K = _quiet(R.tesseroid.gz, lon, lat, h, mesh)
    b = np.array([c.get_bounds() for c in mesh])
    Ko, info = oracle.tess_gz_kernel(lon, lat, h, b, return_info=True)
    assert _relmax(Ko, K) < 1e-13, _relmax(Ko, K)
    # near-field: small cells right under low observation points (deep subdivision),
    # a thin cell (Lr <= 1e3 -> error code, no radial split) and a polar cap cell
    cells = np.array([[10, 10.5, 20, 20.5, 0, -1000], [10, 10.5, 20, 20.5, -1000, -3000],
                      [10, 11, 20, 21, 2000, -5000], [10, 10.5, 20, 20.5, 0, -500],
                      [-180, -150, 60, 90, 0, -300000], [0, 3, -90, -87, -2.7e6, -3e6]], float)
    o_lon = np.array([10.25, 10.25, 10.0, 10.6, 12.0, 10.25, -170.0, 0.0, 100.0])
    o_lat = np.array([20.25, 20.25, 20.0, 20.4, 22.0, 20.3, 89.0, -90.0, -45.0])
    o_h = np.array([100.0, 2500.0, 50.0, 1000.0, 5000.0, 10.0, 5000.0, 5000.0, 250000.0])
    import warnings
    Kn = np.zeros((o_lon.size, len(cells)))
    errs = np.zeros(len(cells), dtype=np.int64)
    lonr, sinlat, coslat, radius = R.tesseroid._convert_coords(o_lon, o_lat, o_h)
    for c, bb in enumerate(cells):
        stack = np.empty((100, 6))
        res = np.zeros(o_lon.size)
        k2 = np.zeros((o_lon.size, len(cells)))
        errs[c] = R.tesseroid._tesseroid_numba.gz(
            lonr, sinlat, coslat, radius, np.array(bb), 1.0, 1.6, stack, np.empty(2),
            np.empty(2), np.empty(2), np.empty(2), res, k2, c)
        Kn[:, c] = k2[:, c]
    Kn = Kn * R.constants.SI2MGAL * R.constants.G
    Kno, info_n = oracle.tess_gz_kernel(o_lon, o_lat, o_h, cells, return_info=True)
    assert _relmax(Kno, Kn) < 1e-13, _relmax(Kno, Kn)
    assert info_n["err_cells"] == int((errs != 0).sum()), (info_n, errs)
    np.savez_compressed(os.path.join(GOLD, "tess_cases.npz"), lon=lon, lat=lat, h=h, bounds=b,
                        K=K, gz=gz, rho=rho, leaves=info["leaves"],
                        n_cells=cells, n_lon=o_lon, n_lat=o_lat, n_h=o_h, n_K=Kn, n_err=errs,
                        n_leaves=info_n["leaves"])
    print("tess_cases", K.shape, "errs", errs, "leaves", info["leaves"], info_n["leaves"])


def _small_problem(R, fixed=False):
    mrange, mspacing = (0, 2000, 0, 3000, 0, 1000), (250, 500, 400)
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 3000, 7), np.linspace(0, 2000, 6))]
    zp = np.zeros_like(xp)
    rng = np.random.default_rng(0)
    dobs = rng.normal(size=xp.size)
    gfix = rng.normal(size=xp.size) * 0.3
    gm = _quiet(R.potential.GravMagModule, dobs, mrange, mspacing, (xp, yp, zp),
                fixed=fixed, grav_fix=gfix if fixed else [])
    return gm, (xp, yp, zp), dobs, gfix


def potential_small(R):
    out = {}
    gm, (xp, yp, zp), dobs, gfix = _small_problem(R)
    gmf, _, _, _ = _small_problem(R, fixed=True)
    wm = gm.Wm.diagonal()
    M = wm.size
    rng = np.random.default_rng(1)
    xs = np.stack([0.001 * wm, rng.uniform(0, 1, M) * wm, rng.uniform(-1, 1, M) * wm])
    mwapr = 0.001 * wm
    out.update(xp=xp, yp=yp, zp=zp, dobs=dobs, gfix=gfix, Aw=np.asfortranarray(gm.Aw), wm=wm,
               xs=xs, mwapr=mwapr, shape=np.array(gm.mshape), mrange=np.array(gm.mrange, float),
               mspacing=np.array(gm.mspacing, float), alpha=0.7, beta=0.001)
    for reg in ("Damping", "MS", "Smoothness", "TV"):
        for tag, g in (("", gm), ("_fix", gmf)):
            res = [g.misfit_and_grad(x, mwapr, None, None, 'mandatory', 1000, 0.7,
                                     regulization=reg, beta=0.001) for x in xs]
            P = oracle.Problem(gm.Aw, dobs, mwapr, reg, 0.7, 0.001, wm=wm, shape=gm.mshape,
                               grav_fix=gfix if tag else None)
            for x, r in zip(xs, res):
                o = P.misfit_and_grad(x)
                assert abs(o[0] - r[0]) <= 1e-13 * abs(r[0]), (reg, tag)
                assert _relmax(o[1], r[1]) < 1e-13 and _relmax(o[2], r[2]) < 1e-13
            out["%s%s_misfit" % (reg, tag)] = np.array([r[0] for r in res])
            out["%s%s_grad" % (reg, tag)] = np.stack([r[1] for r in res])
            out["%s%s_dpre" % (reg, tag)] = np.stack([r[2] for r in res])
            out["%s%s_data" % (reg, tag)] = np.array([r[3] for r in res])
            out["%s%s_model" % (reg, tag)] = np.array([r[4] for r in res])
    np.savez_compressed(os.path.join(GOLD, "potential_small.npz"), **out)
    print("potential_small", gm.Aw.shape)


def _make_chain(R, gm, dt, Lrange, Sigma, alpha, reg, beta, low, high, init, prior, dobs,
                seed, folder):
    """Build the reference's sampler object exactly as hmc.HMCSample does (hmc.py:358-403)."""
    from scipy.sparse import coo_matrix
    chain = R.hmc.HamitonianMC(gm)
    chain.myrank, chain.save_folder, chain.seed = 0, folder, seed
    M = init.size
    chain.boundaries = np.c_[low, high]
    chain.constraint, chain.log_factor = "mandatory", 1000
    chain.Lrange, chain.dt, chain.Sigma = Lrange, dt, Sigma
    chain.adaptiveRegul, chain.RegulRate, chain.RegulFactor = "Fixed", 0.8, alpha
    chain.regularization, chain.beta = reg, beta
    row = np.arange(M)
    chain.invert_Mass = coo_matrix((np.ones(M), (row, row))).tocsr()
    _, _, Wm = chain._kernelw()
    chain.low, chain.high = Wm @ low, Wm @ high
    chain.im = [0, 0]
    chain.initial_model, chain.aprior_model = Wm @ init, Wm @ prior
    chain.dobs, chain.plotsamples = dobs, False
    return chain


def leapfrog_small(R):
    """Single trajectories through the reference's _leapfrog with the RNG stream replayed."""
    gm, _, dobs, _ = _small_problem(R)
    wm = gm.Wm.diagonal()
    M = wm.size
    recs = []
    cases = [("Damping", 0.01, 0.001, 0.0, 1.0, 100), ("MS", 0.01, 0.001, 0.0, 1.0, 101),
             ("TV", 0.02, 0.05, 0.0, 0.05, 102), ("Smoothness", 0.5, 0.5, 0.0, 1.0, 103),
             ("Damping", 0.3, 1.0, 0.0, 0.02, 104), ("MS", 0.6, 2.0, -0.01, 0.01, 105)]
    P_pin = {}
    for reg, dt, Sigma, lo, hi, seed in cases:
        chain = _make_chain(R, gm, dt, [5, 20], Sigma, 1.0, reg, 0.001, np.full(M, lo),
                            np.full(M, hi), np.full(M, 0.5 * (lo + hi) + 0.001), np.full(M, 0.001),
                            dobs, seed, "/tmp/_gold_chain")
        np.random.seed(seed)
        x = chain.initial_model.copy()
        for it in range(4):
            state = np.random.get_state()
            L = np.random.randint(5, 21)
            p0 = np.random.randn(M) * Sigma
            u_peek_state = np.random.get_state()
            u = np.random.rand()
            np.random.set_state(state)
            L2 = np.random.randint(5, 21)
            assert L2 == L
            x_in = x.copy()
            x_out, U, dsyn, acc, Ud, Um = chain._leapfrog(x.copy(), dt, L, 1.0, it)
            # stream position must now be just after rand()
            np.random.set_state(u_peek_state)
            assert np.random.rand() == u
            P = oracle.Problem(gm.Aw, dobs, chain.aprior_model, reg, 1.0, 0.001, wm=wm,
                               shape=gm.mshape)
            xo, acco, out, dso = P.leapfrog(x_in, p0, dt, L, chain.low, chain.high, u)
            assert acco == bool(acc), (reg, it)
            assert _relmax(xo, x_out) < 1e-12, (reg, it, _relmax(xo, x_out))
            assert abs(out[0] - U) <= 1e-12 * abs(U)
            recs.append(dict(reg=reg, dt=dt, L=L, p0=p0, u=u, x_in=x_in, x_out=np.array(x_out),
                             acc=bool(acc), U=U, Ud=Ud, Um=Um, dsyn=np.array(dsyn),
                             low=chain.low.copy(), high=chain.high.copy(),
                             mwapr=chain.aprior_model.copy()))
            x = np.array(x_out)
    out = {"n": len(recs)}
    for i, r in enumerate(recs):
        for k, v in r.items():
            out["%d_%s" % (i, k)] = v
    np.savez_compressed(os.path.join(GOLD, "leapfrog_small.npz"), **out)
    print("leapfrog_small", len(recs), "accepted:", [r["acc"] for r in recs],
          "clamped:", [bool(((r["x_out"] == r["low"]) | (r["x_out"] == r["high"])).any())
                       for r in recs])


def chain_small(R):
    """Whole reference HMCSample runs on the small problem: stdout lines + sample files."""
    gm, _, dobs, _ = _small_problem(R)
    M = gm.Wm.shape[0]
    out = {}
    import shutil
    for tag, reg, dt, Sigma, lo, hi, n in (("a", "Damping", 0.01, 0.001, 0.0, 1.0, 12),
                                            ("b", "TV", 0.02, 0.3, 0.0, 0.02, 12)):
        folder = "/tmp/_gold_hmc_%s_chain" % tag
        shutil.rmtree(folder + "0", ignore_errors=True)
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            R.hmc.HMCSample(gm, n, 0, dt, [5, 20], np.full(M, 0.001 + lo), np.full(M, 0.001),
                            np.c_[np.full(M, lo), np.full(M, hi)], "mandatory", 1000, dobs,
                            "Fixed", 0.8, 1.0, reg, 0.001, 100, Sigma, nbest=100, myrank=0,
                            save_folder=folder, plotsamples=False, im=[0, 0])
        lines = [l for l in buf.getvalue().splitlines() if l.startswith("chain ")]
        out[tag + "_lines"] = np.array(lines)
        out[tag + "_misfit"] = np.loadtxt(folder + "0/misfit.dat")
        out[tag + "_model"] = np.loadtxt(folder + "0/model.dat")
        out[tag + "_misfit_txt"] = np.array(open(folder + "0/misfit.dat").read())
        out[tag + "_model_txt"] = np.array(open(folder + "0/model.dat").read())
        out[tag + "_cfg"] = np.array([dt, Sigma, lo, hi, n])
        out[tag + "_reg"] = np.array(reg)
        print("chain_small", tag, len(lines), "lines; last:", lines[-1])
    np.savez_compressed(os.path.join(GOLD, "chain_small.npz"), **out)


def _chain_lines(path, rank, nmax):
    pat = re.compile(r"^chain %d: .*misfit\(total, data, alpha, model\)=\(([-\d.]+),([-\d.]+),"
                     r"([-\d.]+),([-\d.]+)\) -- accept ratio ([\d.]+)%%" % rank)
    rows = []
    for line in open(path, encoding="utf-8", errors="replace"):
        m = pat.match(line)
        if m:
            rows.append([float(g) for g in m.groups()])
            if len(rows) >= nmax:
                break
    return np.array(rows)


def example_inputs(R):
    """Reference example INPUT DATA files + the known-answer lines of the committed logs."""
    ex = os.path.join(REF, "example")
    out = {}
    # realdata (tesseroid, carved segment mesh, dense, Damping, fixed cells)
    d = os.path.join(ex, "realdata", "data")
    out["real_obs"] = np.loadtxt(os.path.join(d, "gravinv_12d05d.dat"), usecols=[0, 1, 2, 3])
    out["real_gravsea"] = np.loadtxt(os.path.join(d, "grasea_12d05d.dat"), usecols=[2])
    out["real_topo"] = np.loadtxt(os.path.join(d, "topo_12d05d.dat"), usecols=[0, 1, 2])
    out["real_aprior"] = np.loadtxt(os.path.join(d, "SC_ApriorModel.txt"), usecols=[3])
    log = os.path.join(ex, "realdata", "logout_T0.txt")
    out["real_T0_chain0"] = _chain_lines(log, 0, 40)
    out["real_T0_chain1"] = _chain_lines(log, 1, 40)
    out["real_T0_initial_mw_head"] = np.array([0.10433689, 0.15531027, 0.18889759])
    out["real_T0_initial_mw_tail"] = np.array([0.34236374, 0.33920464, 0.32867859])
    # uniformgrid / segmentgrid (prism, wavelet 3D + MS): obs files + log lines
    out["uni_obs"] = np.loadtxt(os.path.join(ex, "uniformgrid", "modeldata",
                                             "model01_singlecube_gz_noise.txt"))
    out["uni_rho"] = np.loadtxt(os.path.join(ex, "uniformgrid", "modeldata",
                                             "model01_singlecube_rho.dat"))
    log = os.path.join(ex, "uniformgrid", "logout_T1.txt")
    out["uni_T1_chain0"] = _chain_lines(log, 0, 40)
    out["uni_T1_chain1"] = _chain_lines(log, 1, 40)
    out["uni_initial_mw_head"] = np.array([1.11688406e-03, 1.17997832e-03, 1.22696931e-03])
    out["uni_initial_mw_tail"] = np.array([5.43033442e-05, 5.17382029e-05, 4.86953394e-05])
    segdir = os.path.join(ex, "segmentgrid", "modeldata")
    segobs = [f for f in os.listdir(segdir) if f.endswith("_gz_noise.txt")]
    out["seg_obs"] = np.loadtxt(os.path.join(segdir, segobs[0]))
    log = os.path.join(ex, "segmentgrid", "logout_T0.txt")
    out["seg_T0_chain0"] = _chain_lines(log, 0, 40)
    out["seg_T0_chain1"] = _chain_lines(log, 1, 40)
    # ratiogrid: geometric dz (ratio 1.05), shape (19, 30, 30): ODD lengths on every wavelet level
    out["ratio_obs"] = np.loadtxt(os.path.join(ex, "ratiogrid", "modeldata", "model_ratio_gz_noise.txt"))
    log = os.path.join(ex, "ratiogrid", "logout_T1.txt")
    out["ratio_T1_chain0"] = _chain_lines(log, 0, 40)
    out["ratio_T1_chain1"] = _chain_lines(log, 1, 40)
    out["ratio_initial_mw_head"] = np.array([2.28807790e-03, 2.36481323e-03, 2.42123838e-03])
    out["ratio_initial_mw_tail"] = np.array([5.19145835e-05, 5.08185019e-05, 4.96606275e-05])
    for k in ("real_T0_chain0", "uni_T1_chain0", "seg_T0_chain0", "ratio_T1_chain0"):
        print(k, out[k].shape, out[k][0])
    np.savez_compressed(os.path.join(GOLD, "example_inputs.npz"), **out)


def c1_leapfrog_rows(R, xp, yp, zp):
    """first rows of misfit.dat for C1 + Damping, dense G (reference run)."""
    ex = os.path.join(REF, "example", "uniformgrid", "modeldata")
    dobs = np.loadtxt(os.path.join(ex, "model01_singlecube_gz_noise.txt"), usecols=[3])
    gm = _quiet(R.potential.GravMagModule, dobs, (0, 2000, 0, 3000, 0, 1000), (100, 100, 100),
                (xp, yp, zp))
    M = 6000
    import shutil
    folder = "/tmp/_gold_c1_chain"
    shutil.rmtree(folder + "0", ignore_errors=True)
    _quiet(R.hmc.HMCSample, gm, 5, 0, 0.01, [5, 20], np.full(M, 0.001), np.full(M, 0.001),
           np.c_[np.zeros(M), np.ones(M)], "mandatory", 1000, dobs, "Fixed", 0.8, 1.0,
           "Damping", 0.001, 100, 0.001, nbest=100, myrank=0, save_folder=folder)
    rows = np.loadtxt(folder + "0/misfit.dat")
    model = np.loadtxt(folder + "0/model.dat")
    print("c1 misfit rows\n", rows[:, :3])
    np.savez_compressed(os.path.join(GOLD, "c1_chain.npz"), misfit=rows, model_last=model[-1],
                        dobs=dobs)


def mesher_cases(R):
    """Cell-bounds tables, masks and node coordinates of the reference's mesh classes."""
    d = os.path.join(REF, "example", "realdata", "data")
    lt, la, tp = np.loadtxt(os.path.join(d, "topo_12d05d.dat"), usecols=[0, 1, 2], unpack=True)
    rng = np.random.default_rng(1)
    tx, ty, th = rng.uniform(-100, 2100, 400), rng.uniform(-100, 3100, 400), -rng.uniform(0, 400, 400)
    cases = {
        "uniform": ("PrismMesh", ((0, 2000, 0, 3000, 0, 1000), (100, 100, 100)), None),
        "uniform_odd": ("PrismMesh", ((0, 2050, 0, 3010, 0, 990), (130, 170, 110)), None),
        "ratio": ("PrismMesh", ((0, 3000, 0, 3000, 0, 2000), (50, 100, 100), 1.2), None),
        "segment": ("PrismMeshSegment", ((0, 2000, 0, 3000, 0, 2100), ([100, 200, 300], 100, 100),
                                         [0, 300, 900, 2100]), None),
        "tess": ("TesseroidMesh", ((-180, 180, -90, 90, 0, -3e6), (-300000, 30, 30)), None),
        "global": ("TesseroidMesh", ((-180, 180, -90, 90, 0, -3000000), (-300000, 3, 3)), None),
        "realdata": ("TesseroidMeshSegment", ((106.5, 118.5, 16, 28, 2000, -60000),
                                              ([-1000, -2000, -5000], 0.5, 0.5),
                                              [2000, -5000, -15000, -60000]), (lt, la, tp)),
        "carve_cubic": ("PrismMesh", ((0, 2000, 0, 3000, 0, 1000), (100, 100, 100)), (tx, ty, th)),
    }
    out = {"names": np.array(list(cases))}
    for name, (cls, args, topo) in cases.items():
        m = _quiet(getattr(R.mesher, cls), *args)
        if topo is not None:
            m.carvetopo(*topo)
        if name == "global":  # 72000 cells: store a strided sample only
            idx = np.arange(0, m.size, 97)
            b = np.array([m[int(i)].get_bounds() for i in idx])
            out[name + "_idx"] = idx
        else:
            b = np.array([c.get_bounds() for c in m if c is not None])
        out[name + "_bounds"] = b
        out[name + "_mask"] = np.array(m.mask, dtype=np.int64)
        out[name + "_shape"] = np.array(m.shape)
        out[name + "_xs"], out[name + "_ys"], out[name + "_zs"] = m.get_xs(), m.get_ys(), m.get_zs()
        if topo is not None:
            out[name + "_topo"] = np.array(topo)
    np.savez_compressed(os.path.join(GOLD, "mesher_cases.npz"), **out)
    print("mesher_cases", {k: out[k + "_bounds"].shape for k in cases})


def cg_small(R):
    """The reference's ConjugateGradient.CG (inversion/reginv.py:357-492) on a small prism problem."""
    from inversion import reginv
    from oracle import cg_port
    mrange, mspacing = (0, 2000, 0, 3000, 0, 1000), (250, 500, 400)
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 3000, 7), np.linspace(0, 2000, 6))]
    zp = np.zeros_like(xp)
    pm = _quiet(R.mesher.PrismMesh, mrange, mspacing)
    b = np.array([c.get_bounds() for c in pm])
    K = oracle.prism_gz_kernel(xp, yp, zp, b)
    rho = np.zeros(pm.shape)
    rho[1:3, 2:4, 1:4] = 0.8
    dobs = K @ rho.ravel() + np.random.default_rng(5).normal(size=xp.size) * 0.01
    cgm = _quiet(reginv.ConjugateGradient, dobs, mrange, mspacing, (xp, yp, zp))
    M = pm.size
    out = dict(xp=xp, yp=yp, zp=zp, dobs=dobs, mrange=np.array(mrange, float), mspacing=np.array(mspacing, float),
               shape=np.array(pm.shape), K=K)
    for reg in ("MS", "Damping", "Smoothness", "TV"):
        res = _quiet(cgm.CG, np.full(M, 0.001), np.full(M, 0.001), (0.0, 1.0), regularization=reg,
                     beta=0.01, q=0.9, maxk=8)
        po = cg_port.cg(K, dobs, pm.shape, np.full(M, 0.001), np.full(M, 0.001), (0.0, 1.0), reg, 0.01, 0.9, 8)
        for a_, b_ in zip(res, po):
            assert _relmax(np.asarray(b_, float), np.asarray(a_, float)) < 1e-9, (reg, _relmax(np.asarray(b_, float), np.asarray(a_, float)))
        for name, v in zip(("model", "data", "dmis", "mmis", "alpha"), res):
            out[reg + "_" + name] = np.asarray(v, dtype=float)
        print("cg_small", reg, "iterations", len(res[2]), "final data misfit", res[2][-1])
    np.savez_compressed(os.path.join(GOLD, "cg_small.npz"), **out)


def bs_small(R):
    """The reference's BootStrap.BSCG (inversion/reginv.py:494-755) on the small prism problem."""
    from inversion import reginv
    from oracle import cg_port
    g = np.load(os.path.join(GOLD, "cg_small.npz"))
    mrange, mspacing = tuple(g["mrange"]), tuple(g["mspacing"])
    bs = _quiet(reginv.BootStrap, mrange, mspacing, (g["xp"], g["yp"], g["zp"]), g["dobs"], (0.0, 1.0),
                samples=3, beta=0.1, maxk=5)
    M = bs.msize
    res = _quiet(bs.BSCG, np.full(M, 0.001))
    po = cg_port.bootstrap(g["K"], g["dobs"], (0.0, 1.0), np.full(M, 0.001), samples=3, beta=0.1, maxk=5)
    for a_, b_ in zip(res, po):
        assert _relmax(b_, a_) < 1e-9, _relmax(b_, a_)
    np.savez_compressed(os.path.join(GOLD, "bs_small.npz"), models=res[0], dmis=res[1], mmis=res[2], alpha=res[3])
    print("bs_small", res[0].shape, res[1][:, -1])


def main():
    os.makedirs(GOLD, exist_ok=True)
    R = ref_harness.load()
    os.chdir("/tmp")
    if len(sys.argv) > 1:
        for name in sys.argv[1:]:
            globals()[name](R)
        return 0
    mesher_cases(R)
    prism_cases(R)
    xp, yp, zp, _ = c1_spot(R)
    tess_cases(R)
    potential_small(R)
    leapfrog_small(R)
    chain_small(R)
    example_inputs(R)
    c1_leapfrog_rows(R, xp, yp, zp)
    cg_small(R)
    bs_small(R)
    print("golden fixtures written to", GOLD)


if __name__ == "__main__":
    sys.exit(main())
