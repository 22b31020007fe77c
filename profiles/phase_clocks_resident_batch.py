"""Per-phase clocks of resident_batch_kernel (GRAVHMC_RESIDENT_TIMING=1) at C1 / C3 with several chains:
    python profiles/phase_clocks_resident_batch.py [chains] [calls] [T]
prints wall time per gh_batch_run call, lock-steps, chain-steps/s and the clocks of workgroup 0 per lock-step."""
import ctypes as C
import os
import sys
import time

import numpy as np

os.environ.setdefault("GRAVHMC_RESIDENT_TIMING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gravinv3dhmc_amd as g  # noqa: E402
from helpers import c1_inputs  # noqa: E402

CH = int(sys.argv[1]) if len(sys.argv) > 1 else 16
CALLS = int(sys.argv[2]) if len(sys.argv) > 2 else 6
T = int(sys.argv[3]) if len(sys.argv) > 3 else 8
REG = sys.argv[4] if len(sys.argv) > 4 else "Damping"
mesh, xp, yp, zp = c1_inputs()
N, M = xp.size, mesh.size
e = g.Engine(N, M)
e.set_obs(xp, yp, zp)
e.set_cells(mesh.cell_bounds(), 0)
e.build_G()
rho = np.zeros(mesh.shape)
rho[2:5, 10:18, 7:11] = 1.0
d = e.forward(rho.ravel())
wm = e.weight(0.5)
e.set_data(d + 0.02 * np.abs(d).max() * np.random.default_rng(0).normal(size=N))
e.set_reg(REG, 1.0, 0.001, mesh.shape, 0.001 * wm)
e.batch_init(np.stack([0.001 * wm] * CH), 0.0 * wm, 1.0 * wm)
rng = np.random.default_rng(1)
e.profile_enable(True)
for call in range(CALLS):
    p0s = rng.normal(size=(CH, T, M)) * 0.001
    Ls = rng.integers(5, 21, size=(CH, T))
    us = rng.uniform(size=(CH, T))
    t0 = time.perf_counter()
    acc, out5, xs, ns, nd = e.batch_run(p0s, 0.01, Ls, us, carry=True)
    dt = time.perf_counter() - t0
    st = e.batch_resident_stats()
    print("call %d: %.2f ms wall, started %s done %s, stats %r" % (call, dt * 1e3, list(ns), list(nd), st), flush=True)
prof = e.profile_read()
st = e.batch_resident_stats()
out = (C.c_longlong * 32)()
la, ev = C.c_int64(0), C.c_int64(0)
e._lib.gh_debug_resident_timing(e._h, out, C.byref(la), C.byref(ev))
names = {0: "update+barrier", 1: "forward+publish", 3: "hop1", 4: "hop2", 6: "hop3", 7: "residual", 11: "adjoint",
         12: "barrier", 14: "decide+update", 10: "end"}
print("kernel: %.3f ms in %d launches, %d lock-steps -> %.2f us per lock-step, %.0f chain-steps/s in the kernel"
      % (prof["sweep_ms"], la.value, ev.value, prof["sweep_ms"] * 1e3 / max(1, ev.value),
         st["chain_steps"] / (prof["sweep_ms"] * 1e-3) if prof["sweep_ms"] else 0.0))
for wgi, base in (("first wg", 0), ("last wg", 16)):
    print(wgi, {names.get(i, str(i)): round(out[base + i] / 100.0 / max(1, ev.value), 3) for i in range(16) if out[base + i]},
          "us per lock-step")
