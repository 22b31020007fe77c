#!/usr/bin/env python3
"""Driver in the style of the reference's example/uniformgrid/main_uniform.py, on the GPU path.

    python examples/run_uniform.py [line index into SetPMTS.txt] [--chains C]
    python -m torch.distributed.run --nproc-per-node K --master-addr 127.0.0.1 examples/run_uniform.py 0

Config lines are the reference's: one Python dict literal per line (parsed with
ast.literal_eval, not eval).  Without a config file the uniformgrid settings are used
(example/uniformgrid/SetPMTS.txt).  Synthetic data: the single-cube model of
example/uniformgrid/model01_singlecube.py forward-modelled on the device.
"""
import argparse
import ast
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gravinv3dhmc_amd as g  # noqa: E402
from gravinv3dhmc_amd import posterior, utils  # noqa: E402
from gravinv3dhmc_amd.dist import Ranks  # noqa: E402

DEFAULT = {"set": "model01_singlecube", "test": "T1", "rhomin": 0, "rhomax": 1,
           "mspacing": [100, 100, 100], "Lrange": [5, 20], "delta": 0.01, "Sigma": 0.001,
           "RegulFactor": 1, "regularization": "MS", "beta": 0.001, "nsamples": 100}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("attempt", nargs="?", type=int, default=0)
    ap.add_argument("--config", default="SetPMTS.txt")
    ap.add_argument("--chains", type=int, default=1, help="chains sharing this GPU (fp64-MFMA batch, or the resident kernel for small problems)")
    ap.add_argument("--wavelet", default=None, choices=[None, "1D", "3D"])
    ap.add_argument("--sink", default="text", choices=["text", "binary", "none"])
    args = ap.parse_args()
    prm = dict(DEFAULT)
    if os.path.exists(args.config):
        with open(args.config) as f:
            prm = [ast.literal_eval(line) for line in f if line.strip()][args.attempt]
    ranks = Ranks()
    mrange = (0, 2000, 0, 3000, 0, 1000)
    nx, ny, nz = 20, 30, 10
    mesh = g.mesher.PrismMesh(mrange, prm["mspacing"])
    rho = np.zeros((nz, ny, nx))
    rho[2:5, 10:18, 7:11] = prm["rhomax"]
    mesh.addprop("density", rho.ravel())
    xp, yp, zp = utils.regular(mrange[:4], (nx, ny), z=0.0)
    gz, _ = g.prism.gz(xp, yp, zp, mesh, return_kernel=False, device=ranks.device)
    dobs = gz + np.random.default_rng(0).normal(0, 0.02 * gz.max(), gz.size)
    start = time.time()
    model = g.GravMagModule(dobs, mrange, prm["mspacing"], (xp, yp, zp), coordinate="cartesian",
                            wavelet=args.wavelet or False, device=ranks.device)
    M = int(np.prod(model.mshape))
    bounds = np.c_[np.full(M, prm["rhomin"]), np.full(M, prm["rhomax"])]
    folder = "result/%s%s_chain" % (prm["set"], prm["test"])
    os.makedirs("result", exist_ok=True)
    common = (prm["delta"], prm["Lrange"], np.full(M, 0.001), np.full(M, 0.001), bounds, "mandatory",
              1000, dobs, "Fixed", 0.8, prm["RegulFactor"], prm["regularization"], prm["beta"], 100,
              prm["Sigma"])
    if args.chains > 1:
        g.HMCSampleBatch(model, args.chains, prm["nsamples"], 0, *common,
                         first_rank=ranks.rank * args.chains, save_folder=folder, sample_sink=args.sink)
    else:
        g.HMCSample(model, prm["nsamples"], 0, *common, myrank=ranks.rank, save_folder=folder,
                    sample_sink=args.sink)
        s = posterior.summarize(model, dobs, rho.ravel())
        print("RMSD:", s["RMSD"])
        print("RMSM:", s["RMSM"])
    ranks.barrier()
    print("total time:", time.time() - start)
    ranks.close()


if __name__ == "__main__":
    main()
