"""Row-block sharding of ONE chain (BASELINE configs[4] as it is worded; SURVEY 8e.2 first form): three ranks on
one GPU over gloo against the CPU oracle.  RCCL over xGMI has only ever run here with world = 1 (1-GPU boxes):
UNMEASURED ON HARDWARE, as the column form."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_row_block_shards_three_ranks_against_the_oracle():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "3",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "shard_worker_rows.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("RESULT ")][-1][len("RESULT "):])
    print("row blocks, 3 ranks:", res)
    assert res["N"] == 40000 and res["rows"] == [13334, 13333, 13333]
    assert res["wm"] < 1e-12 and res["fwd"] < 1e-10 and res["adjoint"] < 1e-10
    for reg in ("MS", "TV"):
        r = res[reg]
        assert r["U"] < 1e-11 and r["grad"] < 1e-10 and r["dpre"] < 1e-10, (reg, r)
    c = res["chain"]
    assert c["decisions_equal"] and c["n"] == 5 and c["out5"] < 1e-9 and c["x"] < 1e-9 and c["accepted"] > 0, c
    # the wavelet-compressed forward on the row blocks (refused on any sharded kernel until round 4) against
    # oracle.Problem(csr=..., dwt=...): the ranks' CSR rows together are the oracle's operator
    w = res["wavelet"]
    assert w["nnz"][0] == w["nnz"][1] and w["ncols"][0] == w["ncols"][1], w
    assert w["U"] < 1e-11 and w["grad"] < 1e-10 and w["dpre"] < 1e-10, w
    c = w["chain"]
    assert c["decisions_equal"] and c["n"] == 5 and c["out5"] < 1e-9 and c["x"] < 1e-9, c
