"""Speed of the library's draws (NumPy's legacy stream, bit for bit) on this host: ns per normal, one thread and with the
scale pass on a second thread behind the generation.
    python profiles/draw_speed.py [M]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from gravinv3dhmc_amd import _lib  # noqa: E402
from gravinv3dhmc_amd.inversion.rng import LegacyDraws  # noqa: E402

lib = _lib.load()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 72000
K = max(2, (24 << 20) // (8 * M))
print("host cores usable:", lib.gh_host_cores(), " block: %d trajectories of %d normals" % (K, M))
ds = {}
for thr in (1, 2):
    d = LegacyDraws(M, (10, 10), 0.001, seed=5)
    lib.gh_rng_set_threads(d._h, thr)
    d.take_block(K)
    ds[thr] = d
res = {t: [] for t in ds}
for rep in range(5):
    for thr, d in ds.items():
        t0 = time.perf_counter()
        d.take_block(K)
        res[thr].append((time.perf_counter() - t0) / (K * M) * 1e9)
for thr in ds:
    print("%2d thread(s): best %.2f ns per normal, %.1f us per trajectory  %s" %
          (thr, min(res[thr]), min(res[thr]) * M / 1e3, [round(v, 2) for v in res[thr]]))
rs = np.random.RandomState(5)
t0 = time.perf_counter()
rs.randn(K * M)
print("numpy randn: %.2f ns per normal" % ((time.perf_counter() - t0) / (K * M) * 1e9))
