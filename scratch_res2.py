import os, sys, time, numpy as np, ctypes as C
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import gravinv3dhmc_amd as G
from gravinv3dhmc_amd import mesher
os.environ["GRAVHMC_RESIDENT"] = "1"; os.environ["GRAVHMC_RESIDENT_TIMING"] = "1"
names = ["dots/prev", "fwd+store", "-", "pollA", "reduce", "pollDone", "pollD", "resid+reg", "pre-E", "E", "tail"]
def run(nx, ny, spacing=(100, 100, 100)):
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 3000, ny), np.linspace(0, 2000, nx))]
    zp = np.zeros_like(xp)
    rng = np.random.default_rng(1)
    dobs = rng.normal(size=xp.size)
    gm = G.GravMagModule(dobs, (0, 2000, 0, 3000, 0, 1000), spacing, (xp, yp, zp), verbose=False)
    eng = gm._engine
    Mc = gm.mesh.size
    wmc = np.asarray(gm.Wm.diagonal())
    eng.set_reg("Damping", 1.0, 0.001, gm.mshape, 0.001 * wmc)
    eng.chain_init(0.001 * wmc, 0 * wmc, 1.0 * wmc)
    tr = [(int(rng.integers(5, 21)), rng.normal(size=Mc) * 0.3, float(rng.uniform())) for _ in range(320)]
    eng.run_chain(iter(tr[:32]), 0.002, lambda *a: None)
    eng.profile_enable(True)
    eng.run_chain(iter(tr), 0.002, lambda *a: None)
    pr = eng.profile_read()
    out = (C.c_longlong * 32)(); la = C.c_int64(0); evs = C.c_int64(0)
    eng._lib.gh_debug_resident_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    eng._lib.gh_debug_resident_timing(eng._h, out, C.byref(la), C.byref(evs))
    print("N=%d M=%d: %.2f us/eval" % (xp.size, Mc, 1e3 * pr["sweep_ms"] / pr["sweeps"]), flush=True)
    for wg in ((0, 1) if evs.value and out[0] else ()):
        print("  wg", "first" if wg == 0 else "last ", " ".join("%s %.2f" % (n, out[16 * wg + i] * 0.01 / evs.value) for i, n in enumerate(names) if n != "-"))
    eng.close()
run(20, 30)
os.environ["GRAVHMC_RESIDENT_TIMING"] = "0"
run(20, 30)
