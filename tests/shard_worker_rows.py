"""Worker of tests/test_gpu_shard_rows.py: three ranks share GPU 0; BASELINE configs[4] in miniature with the
OBSERVATIONS sharded (row blocks, SURVEY 8e.2 first form): the full 200 x 200 observation grid (N = 4*10^4) in
three row blocks of 13334 / 13333 / 13333, 4800 cells of the C5 mesh (replicated model); all-reduces over gloo.
Rank 0 runs the CPU oracle on the same problem and compares."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gravinv3dhmc_amd.dist import Ranks, make_sharded_engine  # noqa: E402  (imports torch first)
from helpers import relmax  # noqa: E402
from shard_worker_c5 import c5_block  # noqa: E402


def main():
    ranks = Ranks()
    ranks.local_rank = 0  # all ranks on GPU 0
    mesh, xp, yp, zp = c5_block()
    N, M = xp.size, mesh.size
    rng = np.random.default_rng(56)
    rho = np.zeros(mesh.shape)
    rho[1:3, 10:30, 15:25] = 1.0
    rho = rho.ravel()
    noise = 0.02 * rng.normal(size=N)
    gfix = 0.01 * rng.normal(size=N)
    x_test = rng.uniform(0, 1, M)
    trajs = [(int(rng.integers(1, 7)), rng.normal(size=M) * 0.02, float(rng.uniform())) for _ in range(5)]
    sh = make_sharded_engine(N, M, ranks, device=0, backend="gloo", axis="rows")
    sh.set_obs(xp, yp, zp)
    sh.set_cells(mesh.cell_bounds(), 0)
    sh.build_G()
    d_true = sh.forward(rho)
    wm = sh.weight(0.5)
    dobs = d_true + noise * np.abs(d_true).max()
    low, high = 0.0 * wm, 0.3 * wm
    got = {}
    for reg, fix in (("MS", None), ("TV", gfix)):
        sh.set_data(dobs, fix)
        sh.set_reg(reg, 0.7, 0.001, mesh.shape, 0.001 * wm)
        got[reg] = sh.misfit_and_grad(x_test * wm)
    r_test = rng.normal(size=N)
    adj = sh.adjoint(r_test)
    sh.set_data(dobs)
    sh.set_reg("MS", 0.7, 0.001, mesh.shape, 0.001 * wm)
    sh.chain_init(0.001 * wm, low, high)
    res = []
    sh.run_chain(iter(trajs), 0.002, lambda L, acc, o, xs: res.append((acc, o.copy(), xs)), want_x=True, batch=2)
    # the wavelet-compressed forward on row blocks (potential.py:693-696; compressor3D.py:17-44 works row by row: every
    # rank compresses its own rows of Aw): potential, gradient and the same chain again
    nnz, ncols = sh.compress_wavelet(3, mesh.shape, 1e-3, 2)
    sh.set_reg("MS", 0.7, 0.001, mesh.shape, 0.001 * wm)
    got_w = sh.misfit_and_grad(x_test * wm)
    sh.chain_init(0.001 * wm, low, high)
    res_w = []
    sh.run_chain(iter(trajs), 0.002, lambda L, acc, o, xs: res_w.append((acc, o.copy(), xs)), want_x=True, batch=2)
    nnzs = ranks.gather(int(nnz))
    rows = ranks.gather(sh.N)
    ranks.barrier()
    if ranks.rank == 0:
        from oracle import oracle as orc
        K = orc.prism_gz_kernel(xp, yp, zp, mesh.cell_bounds())
        out = {"N": int(N), "rows": rows, "fwd": relmax(d_true, K @ rho)}
        Aw, wmo = orc.col_weight(K)
        del K
        out["wm"] = relmax(wm, wmo)
        out["adjoint"] = relmax(adj, Aw.T @ r_test)
        for reg, fix in (("MS", None), ("TV", gfix)):
            Pr = orc.Problem(Aw, dobs, 0.001 * wm, reg, 0.7, 0.001, wm=wm, shape=mesh.shape, grav_fix=fix)
            a, b = got[reg], Pr.misfit_and_grad(x_test * wm)
            out[reg] = {"U": abs(a[0] - b[0]) / abs(b[0]), "grad": relmax(a[1], b[1]), "dpre": relmax(a[2], b[2])}
        Pr = orc.Problem(Aw, dobs, 0.001 * wm, "MS", 0.7, 0.001, wm=wm, shape=mesh.shape)
        xo, ref = 0.001 * wm, []
        for (L, p0, u) in trajs:
            xo, acco, oo, _ = Pr.leapfrog(xo, p0, 0.002, L, low, high, u)
            ref.append((acco, oo.copy(), xo.copy()))
        out["chain"] = {"n": len(res), "decisions_equal": [r[0] for r in res] == [c[0] for c in ref],
                        "out5": max(relmax(r[1], c[1]) for r, c in zip(res, ref)),
                        "x": max([relmax(r[2], c[2]) for r, c in zip(res, ref) if r[0]] or [0.0]),
                        "accepted": int(sum(r[0] for r in res))}
        from oracle import wavelet as ow
        csr = ow.compress_kernel(Aw, 3, mesh.shape)
        Pw = orc.Problem(Aw, dobs, 0.001 * wm, "MS", 0.7, 0.001, wm=wm, shape=mesh.shape, csr=csr,
                         dwt=lambda v: ow.model_coeffs(v, 3, mesh.shape))
        a, b = got_w, Pw.misfit_and_grad(x_test * wm)
        out["wavelet"] = {"nnz": [int(sum(nnzs)), int(csr.nnz)], "ncols": [int(ncols), int(csr.shape[1])],
                          "U": abs(a[0] - b[0]) / abs(b[0]), "grad": relmax(a[1], b[1]), "dpre": relmax(a[2], b[2])}
        xo, ref = 0.001 * wm, []
        for (L, p0, u) in trajs:
            xo, acco, oo, _ = Pw.leapfrog(xo, p0, 0.002, L, low, high, u)
            ref.append((acco, oo.copy(), xo.copy()))
        out["wavelet"]["chain"] = {"n": len(res_w), "decisions_equal": [r[0] for r in res_w] == [c[0] for c in ref],
                                   "out5": max(relmax(r[1], c[1]) for r, c in zip(res_w, ref)),
                                   "x": max([relmax(r[2], c[2]) for r, c in zip(res_w, ref) if r[0]] or [0.0]),
                                   "accepted": int(sum(r[0] for r in res_w))}
        print("RESULT " + json.dumps(out))
    ranks.barrier()
    sh.close()
    ranks.close()


if __name__ == "__main__":
    main()
