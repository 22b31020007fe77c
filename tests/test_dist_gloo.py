"""CPU suite: the N>1 launch path with world_size 2 over gloo (no GPU): rank bookkeeping,
barrier, max/sum/gather, and that two ranks draw the reference's two independent RNG streams
(seed + rank) -- the chains are independent, so there is no data-path collective to test."""
import os
import socket
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from gravinv3dhmc_amd.dist import Ranks
from conftest import gold
from oracle import oracle
r = Ranks()
assert r.world == 2 and r.device == r.local_rank
p = gold("potential_small.npz")
wm = p["wm"]; M = wm.size
# one short chain per rank, seed 100 + rank, trajectories by the CPU oracle (test-only stand-in
# for the device engine: this test is about the launch path, not the kernels)
P = oracle.Problem(p["Aw"], p["dobs"], 0.001 * wm, "Damping", 1.0, 0.001, wm=wm)
np.random.seed(r.chain_seed(100))
x = 0.001 * wm; U = []
for it in range(3):
    L = np.random.randint(5, 21); p0 = np.random.randn(M) * 0.001; u = np.random.rand()
    x, acc, out, _ = P.leapfrog(x, p0, 0.01, L, 0.0 * wm, 1.0 * wm, u)
    U.append(float(out[0]))
r.barrier()
tmax = r.max(1.0 + r.rank)
tsum = r.sum(1.0 + r.rank)
allU = r.gather({"rank": r.rank, "U": U, "folder": r.chain_folder("result/x_chain")})
if r.rank == 0:
    print(json.dumps({"tmax": tmax, "tsum": tsum, "all": allU}))
r.close()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_two_ranks_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "tests")]))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    res = json.loads(line)
    assert res["tmax"] == 2.0 and res["tsum"] == 3.0
    a, b = res["all"]
    assert (a["rank"], b["rank"]) == (0, 1)
    assert a["folder"].endswith("chain0") and b["folder"].endswith("chain1")
    assert a["U"] != b["U"]                       # different seeds -> different chains
    # rank 0 equals a single-process run with seed 100 (chains do not interact)
    from conftest import gold
    from oracle import oracle
    p = gold("potential_small.npz")
    wm = p["wm"]
    P = oracle.Problem(p["Aw"], p["dobs"], 0.001 * wm, "Damping", 1.0, 0.001, wm=wm)
    np.random.seed(100)
    x, U = 0.001 * wm, []
    for it in range(3):
        L = np.random.randint(5, 21)
        p0 = np.random.randn(wm.size) * 0.001
        u = np.random.rand()
        x, acc, o, _ = P.leapfrog(x, p0, 0.01, L, 0.0 * wm, 1.0 * wm, u)
        U.append(float(o[0]))
    assert U == a["U"]


def test_single_process_defaults():
    from gravinv3dhmc_amd.dist import Ranks
    env = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        r = Ranks()
        assert (r.rank, r.world, r.device) == (0, 1, 0)
        assert r.max(3.5) == 3.5 and r.gather("a") == ["a"] and r.chain_seed(100) == 100
        r.barrier()
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v


def test_column_partition_and_allgather_single():
    from gravinv3dhmc_amd.dist import Ranks, allgather_slices, column_partition
    assert column_partition(10, 3) == [(0, 4), (4, 7), (7, 10)]
    assert column_partition(2400000, 8)[-1] == (2100000, 2400000)
    # whole z-planes (Smoothness/TV on a sharded model): C1 over 3 ranks, C5 over 8
    assert column_partition(6000, 3, align=600) == [(0, 2400), (2400, 4200), (4200, 6000)]
    c5 = column_partition(2400000, 8, align=40000)
    assert [(b - a) // 40000 for a, b in c5] == [8, 8, 8, 8, 7, 7, 7, 7] and c5[-1][1] == 2400000
    import pytest
    with pytest.raises(ValueError):
        column_partition(6001, 3, align=600)
    parts = column_partition(7, 1)
    env = {k: os.environ.pop(k, None) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}
    try:
        r = Ranks()
        assert np.array_equal(allgather_slices(r, np.arange(7.0), parts), np.arange(7.0))
        assert np.array_equal(r.allreduce_array(np.arange(3.0)), np.arange(3.0))
        assert r.broadcast_bytes(b"abc") == b"abc"
    finally:
        for k, v in env.items():
            if v is not None:
                os.environ[k] = v


SHARD_WORKER = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
from gravinv3dhmc_amd.dist import Ranks, allgather_slices, column_partition
r = Ranks()
parts = column_partition(11, r.world)
m0, m1 = parts[r.rank]
full = allgather_slices(r, np.arange(m0, m1, dtype=float) * 2.0, parts)
s = r.allreduce_array(np.full(4, 1.0 + r.rank))
raw = r.broadcast_bytes(bytes(range(128)) if r.rank == 0 else b"")
if r.rank == 1:
    print(json.dumps({"full": full.tolist(), "s": s.tolist(), "raw_ok": raw == bytes(range(128))}))
r.close()
'''


def test_shard_helpers_two_ranks(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(SHARD_WORKER % ROOT)
    env = dict(os.environ, PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert res["full"] == [2.0 * i for i in range(11)] and res["s"] == [3.0] * 4 and res["raw_ok"]
