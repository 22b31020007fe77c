"""Worker of tests/test_gpu_parity.py::test_sharded_chain_three_ranks_one_gpu: three ranks share
GPU 0, the cells of one model are split between them in whole z-planes (4 + 3 + 3 of the 10
planes), the all-reduces run over gloo."""
import json
import os
import sys

import numpy as np

# several processes of this test share one GPU: the resident chain kernel wants every CU for itself
# (two of them launched together could wait for each other until their time-out), so the unsharded
# comparison engines use the sweep-per-launch path
os.environ["GRAVHMC_RESIDENT"] = "0"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from gravinv3dhmc_amd.dist import Ranks  # noqa: E402  (imports torch first: one HIP runtime)
import gravinv3dhmc_amd as g  # noqa: E402
from conftest import gold  # noqa: E402
from helpers import c1_inputs, relmax  # noqa: E402


def main():
    ranks = Ranks()
    backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
    ranks.local_rank = 0  # all ranks on GPU 0
    gc = gold("c1_chain.npz")
    mesh, xp, yp, zp = c1_inputs()
    dobs = gc["dobs"]
    M = 6000
    kw = dict(verbose=False)
    sharded = g.GravMagModule(dobs, (0, 2000, 0, 3000, 0, 1000), (100, 100, 100), (xp, yp, zp),
                              shard=ranks, shard_backend=backend, shard_planes=True, **kw)
    single = g.GravMagModule(dobs, (0, 2000, 0, 3000, 0, 1000), (100, 100, 100), (xp, yp, zp), **kw)
    out = {"rank": ranks.rank, "M_local": sharded._engine.M_local}
    wm, wm1 = sharded.Wm.diagonal(), single.Wm.diagonal()
    out["wm"] = relmax(wm, wm1)
    rng = np.random.default_rng(3)
    x = rng.uniform(0, 1, M) * wm1
    out["fwd"] = relmax(sharded._engine.forward(x), single._engine.forward(x))
    r = rng.normal(size=600)
    out["adj"] = relmax(sharded._engine.adjoint(r), single._engine.adjoint(r))
    errs = []
    for reg in ("Damping", "MS", "Smoothness", "TV"):
        a = sharded.misfit_and_grad(x, 0.001 * wm1, None, None, "mandatory", 1000, 0.7, regulization=reg, beta=0.001)
        b = single.misfit_and_grad(x, 0.001 * wm1, None, None, "mandatory", 1000, 0.7, regulization=reg, beta=0.001)
        errs += [abs(a[0] - b[0]) / abs(b[0]), relmax(a[1], b[1]), relmax(a[2], b[2]), abs(a[4] - b[4]) / abs(b[4])]
    out["potential"] = max(errs)
    # cells split without regard to the planes (2000 each): the stencil kinds are refused
    ragged = g.GravMagModule(dobs, (0, 2000, 0, 3000, 0, 1000), (100, 100, 100), (xp, yp, zp),
                             shard=ranks, shard_backend=backend, **kw)
    out["M_ragged"] = ragged._engine.M_local
    try:
        ragged.misfit_and_grad(x, 0.001 * wm1, None, None, "mandatory", 1000, 0.7, regulization="TV")
        out["tv_refused"] = False
    except NotImplementedError:
        out["tv_refused"] = True
    ragged._engine.close()
    # whole chains: same RNG stream on every rank, pipelined trajectories with speculation
    import contextlib
    import io
    import tempfile
    for reg, key in (("Damping", ""), ("TV", "tv_")):
        res = {}
        for tag, model in (("sharded", sharded), ("single", single)):
            folder = tempfile.mkdtemp(prefix="shard_%s_%d_" % (tag, ranks.rank))
            buf = io.StringIO()
            with contextlib.redirect_stdout(buf):
                g.HMCSample(model, 5, 0, 0.01, [5, 20], np.full(M, 0.001), np.full(M, 0.001),
                            np.c_[np.zeros(M), np.ones(M)], "mandatory", 1000, dobs, "Fixed", 0.8, 1.0,
                            reg, 0.001, 100, 0.001, myrank=0, save_folder=folder + "/chain")
            res[tag] = ([l for l in buf.getvalue().splitlines() if l.startswith("chain ")],
                        np.loadtxt(folder + "/chain0/misfit.dat"), np.loadtxt(folder + "/chain0/model.dat"))
        out[key + "lines_equal"] = res["sharded"][0] == res["single"][0]
        out[key + "misfit"] = relmax(res["sharded"][1], res["single"][1])
        out[key + "model"] = float(np.abs(res["sharded"][2] - res["single"][2]).max())
        if reg == "Damping":
            out["ref_rows"] = relmax(res["sharded"][1][:, :3], gc["misfit"][:, :3])
    out["spec"] = sharded._engine.chain_stats()
    allout = ranks.gather(out)
    if ranks.rank == 0:
        print("RESULT " + json.dumps(allout))
    ranks.close()


if __name__ == "__main__":
    main()
