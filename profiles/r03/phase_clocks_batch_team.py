"""Per-phase clocks of batch_team_kernel (csrc/batchteam.hip.h) at a bench.py workload -- the table of DESIGN 4.10.

    GRAVHMC_BATCH_TEAM=1 GRAVHMC_BT_DBG_MEM=<member> GRAVHMC_BT_DBG_WAVE=<wave> \
        python profiles/r03/phase_clocks_batch_team.py [workload] [chains]

Lane 0 of the chosen wave of the chosen member (range 0) accumulates the 100 MHz wall clock per phase
(GRAVHMC_MFB_TIMING); printed per tile of 16 columns, per round of 11 launches.  GRAVHMC_BT_BREAK=1|2|4 switches
parts of the kernel off (wrong results; timing only): requests for G inside the loop / LDS operands of the forward
MFMAs / parking."""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["GRAVHMC_MFB_TIMING"] = "1"
import bench
import gravinv3dhmc_amd as G
wl = sys.argv[1] if len(sys.argv) > 1 else "c2_uniform_100x100x50"
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 16
mesh, xp, yp, zp, rho = bench.make_problem(wl)
b = mesh.cell_bounds()
N, M = xp.size, b.shape[0]
e = G.Engine(N, M)
e.set_obs(xp, yp, zp); e.set_cells(b, 0, 1.6); e.build_G(); e.synchronize()
d = e.forward(rho); wm = e.weight(0.5)
rng = np.random.default_rng(1)
e.set_data(d + rng.normal(size=N) * 0.02 * np.abs(d).max()); e.set_reg("Damping", 1.0, 0.01, mesh.shape, 0.001 * wm)
e.batch_init(np.stack([0.001 * wm] * Cn), 0.0 * wm, 1.0 * wm)
out = (C.c_longlong * 8)()
lib = e._lib
lib.gh_debug_mfb_timing.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
prev = np.zeros(8)
L = 10
for rnd in range(3):
    p0s = rng.normal(size=(Cn, M)) * 0.001
    e.synchronize(); t0 = time.time()
    e.batch_trajectory(p0s, 0.002, np.full(Cn, L), rng.uniform(size=Cn))
    e.synchronize(); dt = time.time() - t0
    lib.gh_debug_mfb_timing(e._h, out)
    v = np.array(list(out), dtype=float) / 100.0   # us
    fs = e.batch_fused_stats()
    print("round", rnd, "%.1f ms = %.2f ms per step" % (dt * 1e3, dt * 1e3 / (L + 1)), fs)
    dv, prev = v - prev, v
    ntile = (M + 15) // 16 / max(1, fs["ranges"]) * (L + 1)
    names = ["requests+park", "adjoint MFMAs+tile requests (issue)", "red+wait+settle xs", "B1 wait", "publish parts", "sums+updates", "forward MFMAs (+tile requests)", "B3 wait"]
    print({n: round(x / ntile, 3) for n, x in zip(names, dv)}, "us per tile; sum", round(dv.sum() / ntile, 2))
