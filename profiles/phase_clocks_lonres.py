"""Per-phase clocks (GRAVHMC_LONSYM_TIMING=1, workgroup 0, thread 0) of the persistent harmonic pass at C4
(csrc/lonres.hip.h):    python profiles/phase_clocks_lonres.py [steps] [trajectories per call, 0 = default] [regulariser]"""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("GRAVHMC_LONSYM_TIMING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gravinv3dhmc_amd as g  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
batch = (int(sys.argv[2]) or None) if len(sys.argv) > 2 else None
reg = sys.argv[3] if len(sys.argv) > 3 else "Damping"
mesh = g.mesher.TesseroidMesh((-180, 180, -90, 90, 0, -3000000), (-300000, 3, 3))
lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, 3.0), np.arange(-90, 91, 3.0), indexing="ij")]
h = np.full_like(lon, 5000.0)
N, M = lon.size, mesh.size
e = g.Engine(N, M)
e.set_shift_invariant(True)
e.set_obs(lon, lat, h)
e.set_cells(mesh.cell_bounds(), 1, 1.6)
e.build_G()
rho = np.zeros(mesh.shape)
rho[1:4, 20:30, 40:60] = 0.3
d = e.forward(rho.ravel())
wm = e.weight(0.5)
e.set_data(d + 0.02 * np.abs(d).max() * np.random.default_rng(0).normal(size=N))
e.set_reg(reg, 0.05, 0.01, mesh.shape, 0.001 * wm)
e.chain_init(0.001 * wm, 0.0 * wm, 0.8 * wm)
rng = np.random.default_rng(1)
L = 10
trajs = [(L, rng.normal(size=M) * 0.001, float(rng.uniform())) for _ in range(steps // L)]
e.run_chain(iter(trajs[:20]), 0.005, lambda *a: None, batch=batch)
e.synchronize()
NAMES = ["prologue", "forward", "publish", "flags", "hop1", "hop2 (class owner)", "hop3", "adjoint + gradient", "update / decision",
         "next trajectory", "(hop2: cluster sums arrive)", "(hop2: inverse transform, sums)", "-", "(adjoint: products)",
         "(adjoint: quarter sums)", "-"]
out0 = (C.c_longlong * 16)()
e._lib.gh_debug_lonres_timing(e._h, out0)
s0 = e.shift_invariant_resident_stats()
e.profile_enable(True)
t0 = time.perf_counter()
e.run_chain(iter(trajs), 0.005, lambda *a: None, batch=batch)
e.synchronize()
el = time.perf_counter() - t0
prof = e.profile_read()
out = (C.c_longlong * 16)()
e._lib.gh_debug_lonres_timing(e._h, out)
s1 = e.shift_invariant_resident_stats()
ev = s1["evaluations"] - s0["evaluations"]
print("%d steps in %.3f s: %.0f steps/s, %.1f us per step; %d launches, %d evaluations" %
      (steps, el, steps / el, el / steps * 1e6, s1["launches"] - s0["launches"], ev))
print("kernel: %.3f ms -> %.2f us per evaluation" % (prof["sweep_ms"], prof["sweep_ms"] * 1e3 / max(1, prof["sweeps"])))
print("clocks of workgroup 0 per evaluation (us):",
      {NAMES[i]: round((out[i] - out0[i]) / 100.0 / ev, 2) for i in range(len(NAMES))})
