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
    if not (len(sys.argv) > 2 and sys.argv[2] == "one-gpu-per-rank"):
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
    # the regulariser is re-sent only when its VALUES change, decided identically on every rank
    # (gh_set_reg is collective for the stencil kinds): fresh temporaries of equal content, whose
    # addresses differ from rank to rank and call to call, must neither desynchronise the ranks nor
    # trigger a re-send; an in-place edit must
    eng = sharded._engine
    sharded._resend_count = 0
    orig = eng.set_reg

    def counting(*a, **k):
        sharded._resend_count += 1
        return orig(*a, **k)

    eng.set_reg = counting
    prior = 0.001 * wm1
    for k in range(3):
        tmp = np.array(prior) + 0.0                          # fresh temporary every call
        a = sharded.misfit_and_grad(x, tmp, None, None, "mandatory", 1000, 0.7, regulization="TV", beta=0.001)
        b = single.misfit_and_grad(x, prior * 1.0, None, None, "mandatory", 1000, 0.7, regulization="TV", beta=0.001)
        errs.append(abs(a[0] - b[0]) / abs(b[0]))
    out["resends_equal_content"] = sharded._resend_count      # 0: TV with this prior was the last one sent
    prior2 = prior.copy()
    prior2[ranks.world] *= 1.5                                 # same edit on every rank
    a = sharded.misfit_and_grad(x, prior2, None, None, "mandatory", 1000, 0.7, regulization="TV", beta=0.001)
    b = single.misfit_and_grad(x, prior2, None, None, "mandatory", 1000, 0.7, regulization="TV", beta=0.001)
    out["resends_after_edit"] = sharded._resend_count
    out["edit_potential"] = abs(a[0] - b[0]) / abs(b[0])
    eng.set_reg = orig
    out["potential"] = max(errs)
    # more than three batches of two trajectories through run_chain with overlap requested: with the
    # host-staged (gloo) all-reduce no batch may run while results are gathered (one process group,
    # two threads); same chain as the unsharded engine
    rngc = np.random.default_rng(11)
    trajs = [(int(rngc.integers(2, 7)), rngc.normal(size=M) * 0.001, float(rngc.uniform())) for _ in range(9)]
    got = {}
    for tag, model in (("sharded", sharded), ("single", single)):
        e = model._engine
        e.set_reg("MS", 1.0, 0.001, model.mshape, 0.001 * wm1)
        e.chain_init(0.001 * wm1, 0.0 * wm1, 1.0 * wm1)
        res = []
        e.run_chain(iter(trajs), 0.01, lambda L, acc, o, xs, res=res: res.append((acc, o.copy(), xs)), want_x=True,
                    batch=2, overlap=True)
        got[tag] = res
    out["overlap_n"] = len(got["sharded"])
    out["overlap_decisions"] = [r[0] for r in got["sharded"]] == [r[0] for r in got["single"]]
    out["overlap_out5"] = max(relmax(a[1], b[1]) for a, b in zip(got["sharded"], got["single"]))
    out["overlap_x"] = max([relmax(a[2], b[2]) for a, b in zip(got["sharded"], got["single"]) if a[0]] or [0.0])
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
