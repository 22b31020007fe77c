#!/usr/bin/env python3
"""Headline benchmark: HMC leapfrog steps/s + achieved HBM GB/s of the G sweep.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one leapfrog step of the hot path (hmc.py:117-152): momentum update with the
gradient 2 Aw^T r + alpha grad R, position update with clamp-and-reflect, and the forward
product Aw x for the next step -- ONE fused sweep of the resident kernel matrix -- plus the
per-trajectory work that surrounds it (momentum draw and upload, the extra adjoint-only
sweep, the Metropolis test).  Workload at N=1: BASELINE.json configs[1] (uniformgrid
100x100x50 prisms, N=10^4 observations, dense G = 40 GB fp64 resident in HBM, Damping).
With N>1 every rank runs an independent chain on its own GPU against its own copy of G
(seed = 100 + rank: the reference's `mpiexec -n K` model, hmc.py:367-369): no data-path
collective, weak scaling; value = steps of all ranks / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (nx, ny, nz, anomaly block (ix0, ix1, iy0, iy1, iz0, iz1), dt)  -- SURVEY 8d
    # dt: the reference's 0.01 (uniformgrid/SetPMTS.txt) is stable at C1; at C2 the potential
    # is ~80x stiffer (M/N grows), 0.002 keeps the acceptance at the reference's ~100 %.
    "c2_uniform_100x100x50": (100, 100, 50, (40, 59, 40, 59, 10, 24), 0.002),
    "c1_uniform_20x30x10": (20, 30, 10, (7, 10, 10, 17, 2, 4), 0.01),
}


#: fp64 vector peak of MI355X: 256 CUs x 4 SIMDs x 16 fp64 FMA lanes/clk x 2 flop x 2.4 GHz (half the
#: FP32 vector rate of MI355X_MICROARCH.md's table, AMD's datasheet figure)
FP64_VECTOR_PEAK_TFLOPS = 78.6
#: operation counts of the reference's formulas as written
FLOPS_PER_TESS_LEAF = 244      # _tesseroid_numba.py:94-111 (49) + :135-157 (3) + :75-91,207-222 (192)
FLOPS_PER_PRISM_ENTRY = 194    # _prism.pyx:265-290: 8 corners x (3 + 6 + 4 + 3 + 6 + 2) + 2
FLOPS_PER_TESS_PAIR_HOISTED = 146  # per pair and step once the cell-only part is tabulated: 2 x (sub, cos)
                                   # + 4 cospsi x 4 + 8 nodes x 15 + r^2 + sign, scale, sum, 2 unit factors


def pmc_traffic(workload):
    """HBM bytes per sweep launch from the newest committed rocprofv3 PMC summary of this workload
    (profiles/rNN/<tag>_pmc_summary.json: separate --pmc FETCH_SIZE / WRITE_SIZE passes of this
    command, corrected as MI355X_MICROARCH.md prescribes).  Hardware counters cannot be read from
    inside the benchmark process, so this is NOT measured in the run that prints it: returns
    (bytes, {"file", "note"}) for the line's `traffic` / `traffic_from_profile` fields, or
    (None, None) when no profile of this workload exists."""
    import glob
    tag = workload.split("_")[0]
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", tag + "*_pmc_summary.json")),
                    reverse=True):
        try:
            doc = json.load(open(f))
            ks = doc["kernels"]
            # the dominant kernel of the workload: the team sweep where it runs, else the sweep
            k = ([v for n, v in ks.items() if "teamsweep_kernel" in n] or
                 [v for n, v in ks.items() if "sweep_kernel<" in n])[0]
            b = k["hbm_read_bytes"] + k.get("hbm_write_bytes", 0.0)
            return b, {"file": os.path.relpath(f, ROOT), "bytes_per_launch": b,
                       "note": "separate rocprofv3 --pmc passes of this command, committed with the "
                               "sources (commit %s); not a counter of this run" % doc.get("commit", "n/a")}
        except Exception:
            continue
    return None, None


def usable_cores():
    """Host cores this process may really use: the affinity mask capped by the cgroup CPU quota
    (a GPU box hands a container 16 of its 64+ cores: more BLAS/OpenMP threads than that are
    throttled, which made 64 threads look no faster than one in round 1)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
            break
        except Exception:
            continue
    return max(1, n)


def make_problem(name):
    from gravinv3dhmc_amd import mesher
    nx, ny, nz, blk, _dt = WORKLOADS[name]
    mesh = mesher.PrismMesh((0, 100.0 * nx, 0, 100.0 * ny, 0, 100.0 * nz), (100, 100, 100))
    yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 100.0 * ny, ny),
                                             np.linspace(0, 100.0 * nx, nx))]
    zp = np.zeros_like(xp)
    rho = np.zeros((nz, ny, nx))
    rho[blk[4]:blk[5] + 1, blk[2]:blk[3] + 1, blk[0]:blk[1] + 1] = 1.0
    return mesh, xp, yp, zp, rho.ravel()


#: the other BASELINE.json configs, runnable with --workload (parity-test cases, not bench lines)
EXTRA = {
    # C3: segmentgrid, wavelet='3D' compressed forward + dense adjoint, TV
    "c3_segment_wavelet3d_tv": dict(kind=0, reg="TV", alpha=1.0, beta=0.001, dt=0.01, hi=1.0,
                                    wavelet=3),
    # C4: global 3 deg tesseroid mesh, 121 x 61 obs at 5000 m, Damping 0.05, bounds [0, 0.8]
    "c4_global_tesseroid": dict(kind=1, reg="Damping", alpha=0.05, beta=0.01, dt=0.005, hi=0.8,
                                wavelet=0),
    # C5: 200 x 200 x 60 prisms (M = 2.4e6), 200 x 200 obs (N = 4e4): G = 768 GB, one chain sharded
    # over 8 GPUs (--shard).  On one GPU use --cells-fraction 8: the 96 GB share one rank holds.
    "c5_uniform_200x200x60": dict(kind=0, reg="MS", alpha=1.0, beta=0.001, dt=0.001, hi=1.0, wavelet=0),
    # the size of the reference's realdata example (625 x 10427, 52 MB) as prisms: larger than the
    # LDS alone, the resident chain kernel keeps part of every column block in registers
    "x1_prisms_625x10400": dict(kind=0, reg="MS", alpha=1.0, beta=0.001, dt=0.01, hi=1.0, wavelet=0),
    # the reference's ratiogrid example (900 x 17100, dz growing by 1.05, wavelet 3D, MS): 123 MB of Aw
    # + 123 MB of the dense compressed-forward form -- beyond LDS + registers: resident kernel with
    # the columns that do not fit streamed from L2 / Infinity Cache
    "x2_ratiogrid_900x17100_wavelet3d": dict(kind=0, reg="MS", alpha=1.0, beta=0.001, dt=0.01, hi=0.4, wavelet=3),
    # C4's geometry family (example/global/SetPMTS.txt) at 1 degree: 360 x 180 x 10 tesseroids (M = 648 000), 361 x 181
    # observations (N = 65 341) -- a dense kernel of 339 GB; with --shift-invariant the streamed harmonic store, 0.94 GB
    "x3_global_one_degree": dict(kind=1, reg="Damping", alpha=0.05, beta=0.01, dt=0.001, hi=0.8, wavelet=0),
}


def make_extra(name):
    from gravinv3dhmc_amd import mesher
    if name == "c3_segment_wavelet3d_tv":
        mesh = mesher.PrismMeshSegment((0, 2000, 0, 3000, 0, 2100), ([100, 200, 300], 100, 100),
                                       [0, 300, 900, 2100])
        yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 3000, 30), np.linspace(0, 2000, 20))]
        obs = (xp, yp, np.zeros_like(xp))
        rho = np.zeros(mesh.shape)
        rho[2:5, 10:18, 7:11] = 1.0
    elif name == "x1_prisms_625x10400":
        mesh = mesher.PrismMesh((0, 2600, 0, 2000, 0, 2000), (100, 100, 100))
        yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 2000, 25), np.linspace(0, 2600, 25))]
        obs = (xp, yp, np.zeros_like(xp))
        rho = np.zeros(mesh.shape)
        rho[4:9, 7:13, 10:16] = 1.0
    elif name == "x2_ratiogrid_900x17100_wavelet3d":
        mesh = mesher.PrismMesh((0, 6000, 0, 6000, 0, 6000), (200, 200, 200), 1.05)
        yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 6000, 30), np.linspace(0, 6000, 30))]
        obs = (xp, yp, np.zeros_like(xp))
        rho = np.zeros(mesh.shape)
        rho[4:9, 10:20, 10:20] = 0.4
    elif name == "c5_uniform_200x200x60":
        mesh = mesher.PrismMesh((0, 20000, 0, 20000, 0, 6000), (100, 100, 100))
        yp, xp = [a.ravel() for a in np.meshgrid(np.linspace(0, 20000, 200), np.linspace(0, 20000, 200))]
        obs = (xp, yp, np.zeros_like(xp))
        rho = np.zeros(mesh.shape)
        rho[10:30, 80:120, 80:120] = 1.0
    elif name == "x3_global_one_degree":
        mesh = mesher.TesseroidMesh((-180, 180, -90, 90, 0, -3000000), (-300000, 1, 1))
        lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, 1.0), np.arange(-90, 91, 1.0),
                                                   indexing="ij")]
        obs = (lon, lat, np.full_like(lon, 5000.0))
        rho = np.zeros(mesh.shape)
        rho[2:5, 60:90, 100:160] = 0.3
    else:
        mesh = mesher.TesseroidMesh((-180, 180, -90, 90, 0, -3000000), (-300000, 3, 3))
        lon, lat = [a.ravel() for a in np.meshgrid(np.arange(-180, 181, 3.0), np.arange(-90, 91, 3.0),
                                                   indexing="ij")]
        obs = (lon, lat, np.full_like(lon, 5000.0))
        rho = np.zeros(mesh.shape)
        rho[1:4, 20:30, 40:60] = 0.3
    return mesh, obs, rho.ravel()


def cpu_baseline(mesh, xp, yp, zp, dobs, target_s=16.0, max_cells=10000):
    """The reference's CPU formulation of the same step, timed on this box's host cores on a
    bounded sample: same N observations, every k-th cell (cost per step is linear in the number of
    cells: steps/s is scaled by the sampling factor).  Two restatements are timed, each on all
    usable cores and on one: the NumPy + OpenBLAS mirror of potential.py:688-845 / hmc.py:85-177
    (oracle/numpy_port.py: what the reference itself executes) and the C oracle with OpenMP
    (oracle/gravhmc_oracle.c); `value` is the faster.  kind "port": the reference itself cannot
    travel to the GPU box."""
    import ctypes
    from oracle import oracle
    from oracle.numpy_port import NumpyProblem
    from threadpoolctl import threadpool_limits
    cores = usable_cores()
    M = mesh.size
    k = max(1, M // max_cells)
    b = mesh.cell_bounds()[::k]
    t0 = time.time()
    K = oracle.prism_gz_kernel(xp, yp, zp, b)
    Aw, wm = oracle.col_weight(K)
    t_build = time.time() - t0
    del K
    Ms = wm.size
    rng = np.random.default_rng(1)
    L = 10
    low, high = 0.0 * wm, 1.0 * wm
    gomp = None
    for name in ("libgomp.so.1", "libgomp.so"):
        try:
            gomp = ctypes.CDLL(name)
            break
        except OSError:
            continue

    def timed(problem, budget):
        x = 0.001 * wm
        problem.leapfrog(x, rng.normal(size=Ms) * 0.001, 0.01, 2, low, high, 0.5)   # warm-up
        steps, t_run = 0, 0.0
        while t_run < budget and steps < 4000:
            p0 = rng.normal(size=Ms) * 0.001
            t1 = time.time()
            x, _acc, _out, _ = problem.leapfrog(x, p0, 0.01, L, low, high, 0.5)
            t_run += time.time() - t1
            steps += L
        return steps / t_run, steps, t_run

    res = {}
    Pn = NumpyProblem(Aw, dobs, 0.001 * wm, "Damping", 1.0, 0.01, wm=wm)
    Pc = oracle.Problem(Aw, dobs, 0.001 * wm, "Damping", 1.0, 0.01, wm=wm)
    share = target_s / 6.0
    for threads, budget in ((cores, 2 * share), (1, share)):
        with threadpool_limits(limits=threads, user_api="blas"):
            res[("numpy_openblas", threads)] = timed(Pn, budget)
        if gomp is not None:
            gomp.omp_set_num_threads(threads)
        res[("c_openmp", threads)] = timed(Pc, budget)
    if gomp is not None:
        gomp.omp_set_num_threads(cores)
    best = max((("numpy_openblas", cores), ("c_openmp", cores)), key=lambda kk: res[kk][0])
    scale = Ms / M
    table = {"%s_%dthr" % kk: res[kk][0] * scale for kk in sorted(res)}
    return {"value": res[best][0] * scale, "unit": "leapfrog steps/s", "cores": cores,
            "kind": "port", "restatement": best[0],
            "single_core_value": max(res[("numpy_openblas", 1)][0], res[("c_openmp", 1)][0]) * scale,
            "thread_scaling_steps_per_s": table,
            "host_cpus_visible": os.cpu_count(),
            "sample": "same N=%d observations, every %d-th cell (%d of %d, G sample %.0f MB); L=%d; "
                      "%s on %d threads: %d steps in %.1f s (%.2f steps/s on the sample, scaled by %d/%d); "
                      "sample kernel build + weighting %.1f s; cores = affinity capped by the cgroup CPU quota "
                      "(os.cpu_count() = %s)"
                      % (xp.size, k, Ms, M, Aw.nbytes / 1e6, L, best[0], cores, res[best][1], res[best][2],
                         res[best][0], Ms, M, t_build, os.cpu_count())}


#: (U, U_data, U_model) of the chain after its last trajectory: lets two runs of the same workload
#: (e.g. sharded and unsharded) be compared through their JSON lines
LAST_STATE = {"U": None}


def run_single_chain(eng, M, Sigma, dt, L, steps, warmup, seed, barrier=lambda: None):
    """One chain through the sampler's own path (Engine.run_chain), momenta drawn in the reference's
    RNG order (legacy global generator, hmc.py:95,164).  Returns (elapsed_s, accepted, trajectories,
    profile).  Every draw happens inside the timed region, overlapped with the GPU where the pipeline
    manages to."""
    np.random.seed(seed)
    from gravinv3dhmc_amd.inversion.rng import LegacyDraws

    def prepare(total_steps):
        plan = [L] * (total_steps // L) + ([total_steps % L] if total_steps % L else [])
        if os.environ.get("GRAVHMC_HOST_RNG", "native") == "numpy":
            def draws():
                for n in plan:
                    yield n, np.random.randn(M) * Sigma, np.random.rand()
            return plan, draws()
        return plan, LegacyDraws(M, (L, L), Sigma, fixed_L=plan)   # the same stream, drawn by the library

    def run(prepared):
        plan, gen = prepared
        stat = {"acc": 0, "traj": 0}

        def on_result(n, acc, out5, x):
            stat["acc"] += int(acc)
            stat["traj"] += 1
            LAST_STATE["U"] = [float(v) for v in out5[:3]]

        try:
            eng.run_chain(gen, dt, on_result, overlap=True)   # as the sampler calls it (inversion/hmc.py)
        finally:
            if hasattr(gen, "release"):
                gen.release()
        return stat["acc"], stat["traj"]

    if warmup > 0:
        run(prepare(warmup))
    prepared = prepare(steps)
    eng.synchronize()
    barrier()
    eng.profile_enable(True)
    t0 = time.perf_counter()
    naccept, ntraj = run(prepared)
    eng.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()
    prof = eng.profile_read()
    eng.profile_enable(False)
    return elapsed, naccept, ntraj, prof


#: the other BASELINE.json configurations, each measured by a child process of the default run
#: (same script, its own workload) and summarised under `extra`, so one driver record carries them
EXTRA_RUNS = [
    ("c2_hmcsample", ["--workload", "c2_uniform_100x100x50", "--hmcsample", "60"]),
    ("c1_uniform_16_chains", ["--workload", "c1_uniform_20x30x10", "--chains-per-gpu", "16", "--steps", "16000",
                              "--warmup", "1600"]),
    ("c4_hmcsample", ["--workload", "c4_global_tesseroid", "--hmcsample", "1000"]),
    ("c1_hmcsample_batch_16_chains", ["--workload", "c1_uniform_20x30x10", "--chains-per-gpu", "16", "--hmcsample-batch", "600"]),
    ("c2_uniform_16_chains", ["--workload", "c2_uniform_100x100x50", "--chains-per-gpu", "16", "--steps", "60",
                              "--warmup", "20"]),
    ("c2_uniform_16_chains_two_reads_of_G", ["--workload", "c2_uniform_100x100x50", "--chains-per-gpu", "16", "--steps", "60",
                                             "--warmup", "20", "--batch-team", "off"]),
    ("c3_segment_wavelet3d_tv", ["--workload", "c3_segment_wavelet3d_tv", "--steps", "20000", "--warmup", "2000"]),
    ("c3_segment_wavelet3d_tv_16_chains", ["--workload", "c3_segment_wavelet3d_tv", "--chains-per-gpu", "16", "--steps", "4000",
                                           "--warmup", "400"]),
    ("c4_global_tesseroid_matrix_free", ["--workload", "c4_global_tesseroid", "--matrix-free", "--steps", "100",
                                         "--warmup", "10"]),
    ("c4_global_tesseroid_dense", ["--workload", "c4_global_tesseroid", "--steps", "2000", "--warmup", "200"]),
    ("c4_global_tesseroid_shift_invariant", ["--workload", "c4_global_tesseroid", "--shift-invariant", "--steps", "16000",
                                             "--warmup", "1600"]),
    ("c4_global_tesseroid_shift_invariant_8_chains", ["--workload", "c4_global_tesseroid", "--shift-invariant",
                                                      "--chains-per-gpu", "8", "--steps", "2000", "--warmup", "200"]),
    ("c4_global_tesseroid_matrix_free_8_chains", ["--workload", "c4_global_tesseroid", "--matrix-free",
                                                  "--chains-per-gpu", "8", "--steps", "100", "--warmup", "20"]),
    ("x3_global_one_degree_shift_invariant", ["--workload", "x3_global_one_degree", "--shift-invariant", "--steps", "2000",
                                              "--warmup", "200"]),
    ("x3_global_one_degree_shift_invariant_8_chains", ["--workload", "x3_global_one_degree", "--shift-invariant",
                                                       "--chains-per-gpu", "8", "--steps", "400", "--warmup", "40"]),
    ("c5_share_of_one_gpu_of_8", ["--workload", "c5_uniform_200x200x60", "--cells-fraction", "8", "--steps", "40",
                                  "--warmup", "10"]),
    # (the same 96 GB as the row block one of 8 GPUs holds: 5000 observations x all 2.4e6 cells, two reads per step)
    ("c5_share_row_blocks", ["--workload", "c5_uniform_200x200x60", "--rows-fraction", "8", "--shard", "--shard-axis", "rows",
                             "--steps", "20", "--warmup", "10"]),
]


def extra_run(device, extra_args):
    """One of EXTRA_RUNS in a child process (the parent has released its GPU context); returns a
    compact summary of the child's JSON line."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-cpu-baseline", "--no-extra"] + extra_args
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    env["LOCAL_RANK"] = str(device)
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=env)
    if out.returncode != 0:
        return {"error": out.stderr[-400:]}
    l = json.loads([x for x in out.stdout.splitlines() if x.startswith("{")][-1])
    if "hmcsample" in l:
        return l["hmcsample"]
    r = l["roofline"]
    keep = ("bound", "achieved", "peak", "unit", "frac", "kernel", "avg_ms", "launches", "us_per_evaluation",
            "near_field_table", "entries_per_s", "flop_model", "fp64_matrix_TFLOPs", "table", "table_read_GBps", "dense_G_equiv_GBps")
    return {"value": l["value"], "unit": l["unit"], "steps": l["steps"], "warmup": l["warmup"],
            "ms_per_step": l["ms_per_step"],
            "config": {k: l["config"].get(k) for k in ("workload", "N_obs", "M_cells", "G_bytes", "regulariser",
                                                        "matrix_free", "shift_invariant", "chains_per_gpu",
                                                        "wavelet_nnz", "dt", "traj_len", "accepted",
                                                        "trajectories")},
            "roofline": {k: r[k] for k in keep if k in r}}


def hmcsample_global_block(device, nsamples=300):
    """The SAMPLER on the reference's own global example (example/global/main_global.py + SetPMTS.txt: 3 degree
    tesseroid mesh, Damping 0.05, Lrange [5, 20], delta 0.005, bounds [0, 0.8]; BASELINE configs[3] runs it one chain
    per GPU): GravMagModule(coordinate="spherical", shift_invariant=True) -> HMCSample, draws in the reference's RNG
    order, misfit.dat and the accepted models written, until `nsamples` proposals have been accepted -- with the binary
    sample sink and with the reference's text rows (0.8 MB of '%.8f' per sample)."""
    import contextlib
    import shutil
    import tempfile
    import gravinv3dhmc_amd as g
    mesh, obs, rho = make_extra("c4_global_tesseroid")
    lon, lat, h = obs
    N, M = lon.size, mesh.size
    t0 = time.time()
    gm = g.GravMagModule(np.zeros(N), (-180, 180, -90, 90, 0, -3000000), (-300000, 3, 3), (lon, lat, h),
                         coordinate="spherical", device=device, verbose=False, shift_invariant=True)
    eng = gm._engine
    wm = gm.Wm.diagonal()
    d_true = eng.forward(wm * rho)
    dobs = d_true + np.random.default_rng(0).normal(0.0, 0.02 * np.abs(d_true).max(), N)
    eng.set_data(dobs)
    gm.dobs = dobs
    out = {"workload": "c4_global_tesseroid (shift-invariant store)", "N_obs": int(N), "M_cells": int(M), "nsamples": nsamples,
           "Lrange": [5, 20], "dt": 0.005, "setup_s": round(time.time() - t0, 2)}
    ones = np.ones(M)
    tmp = tempfile.mkdtemp(prefix="gravhmc_bench_")
    try:
        for sink in ("binary", "text"):
            with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):
                t0 = time.perf_counter()
                chain = g.HMCSample(gm, nsamples, 0, 0.005, [5, 20], 0.01 * ones, 0.001 * ones,
                                    np.c_[0.0 * ones, 0.8 * ones], "mandatory", 1000, dobs, "Fixed", 0.8, 0.05, "Damping",
                                    0.01, 100, 0.001, save_folder=os.path.join(tmp, sink + "_chain"),
                                    sample_sink=sink, posterior_last=0)
                eng.synchronize()
                el = time.perf_counter() - t0
            folder = os.path.join(tmp, sink + "_chain0")
            written = sum(os.path.getsize(os.path.join(folder, f)) for f in os.listdir(folder))
            out[sink + "_sink"] = {"seconds": el, "accepted_samples_per_s": nsamples / el,
                                   "leapfrog_steps": chain.leapfrog_steps, "trajectories": chain.trajectories,
                                   "leapfrog_steps_per_s": chain.leapfrog_steps / el, "bytes_written": int(written)}
        out["persistent_launch"] = eng.shift_invariant_resident_stats()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    eng.close()
    return out


def hmcsample_block(device, workload, nsamples=60):
    """The SAMPLER at the headline workload (BASELINE configs[1]: "4000 samples"): HMCSample as the
    reference's drivers call it (example/uniformgrid/main_uniform.py:17-88, inversion/hmc.py:252-343) --
    Lrange [5, 20], draws in the reference's RNG order, misfit.dat and the accepted models written --
    until `nsamples` proposals have been accepted, once with the binary sample sink and once with the
    reference's text rows (5.5 MB of '%.8f' per sample at C2); next to it Engine.run_chain alone on the
    same model (the rate the headline line reports).  Host loop, formatter, file I/O and RNG overlap are
    all inside these numbers."""
    import contextlib
    import shutil
    import tempfile
    import gravinv3dhmc_amd as g
    mesh, xp, yp, zp, rho = make_problem(workload)
    N, M = xp.size, mesh.size
    nx, ny, nz = WORKLOADS[workload][:3]
    mrange = (0, 100.0 * nx, 0, 100.0 * ny, 0, 100.0 * nz)
    dt = WORKLOADS[workload][4]
    t0 = time.time()
    gm = g.GravMagModule(np.zeros(N), mrange, (100.0, 100.0, 100.0), (xp, yp, zp), device=device, verbose=False)
    eng = gm._engine
    wm = gm.Wm.diagonal()
    d_true = eng.forward(wm * rho)
    dobs = d_true + np.random.default_rng(0).normal(0.0, 0.02 * np.abs(d_true).max(), N)
    eng.set_data(dobs)
    gm.dobs = dobs
    t_setup = time.time() - t0
    out = {"workload": workload, "N_obs": int(N), "M_cells": int(M), "nsamples": nsamples, "Lrange": [5, 20],
           "dt": dt, "setup_s": round(t_setup, 2)}
    ones = np.ones(M)
    tmp = tempfile.mkdtemp(prefix="gravhmc_bench_")
    try:
        for sink in ("binary", "text"):
            with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):
                t0 = time.perf_counter()
                chain = g.HMCSample(gm, nsamples, 0, dt, [5, 20], 0.001 * ones, 0.001 * ones,
                                    np.c_[0.0 * ones, ones], "mandatory", 1000, dobs, "Fixed", 0.8, 1.0, "Damping",
                                    0.01, 100, 0.001, save_folder=os.path.join(tmp, sink + "_chain"),
                                    sample_sink=sink, posterior_last=0)
                eng.synchronize()
                el = time.perf_counter() - t0
            folder = os.path.join(tmp, sink + "_chain0")
            written = sum(os.path.getsize(os.path.join(folder, f)) for f in os.listdir(folder))
            out[sink + "_sink"] = {"seconds": el, "accepted_samples_per_s": nsamples / el,
                                   "leapfrog_steps": chain.leapfrog_steps, "trajectories": chain.trajectories,
                                   "leapfrog_steps_per_s": chain.leapfrog_steps / el,
                                   "bytes_written": int(written)}
        # the chain driver alone (what the headline `value` measures), same model, L = 12
        eng.set_reg("Damping", 1.0, 0.01, mesh.shape, 0.001 * wm)
        eng.chain_init(0.001 * wm, 0.0 * wm, 1.0 * wm)
        steps = out["binary_sink"]["leapfrog_steps"]
        el, nacc, ntraj, prof = run_single_chain(eng, M, 0.001, dt, 12, steps, 24, 100)
        out["run_chain_steps_per_s"] = steps / el
        out["sampler_vs_run_chain"] = out["binary_sink"]["leapfrog_steps_per_s"] / out["run_chain_steps_per_s"]
        # (the difference of two 4.7 s runs is within their noise at C2: never reported below zero)
        out["text_sink_cost_s_per_sample"] = max(0.0, out["text_sink"]["seconds"] - out["binary_sink"]["seconds"]) / nsamples
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    eng.close()
    return out


def hmcsample_batch_block(device, workload, chains, nsamples):
    """The BATCH sampler (HMCSampleBatch: `chains` chains of the reference's ranks first_rank .. on one GPU,
    inversion/hmc.py) on a small workload -- at C1 the chains run in lock-step inside the resident batch kernel --
    until every chain has `nsamples` accepted samples: leapfrog steps/s of all chains with the console lines, the
    misfit rows and the sample sink inside the number (sink "none" and "binary")."""
    import contextlib
    import shutil
    import tempfile
    import gravinv3dhmc_amd as g
    mesh, xp, yp, zp, rho = make_problem(workload)
    N, M = xp.size, mesh.size
    nx, ny, nz = WORKLOADS[workload][:3]
    dt = WORKLOADS[workload][4]
    gm = g.GravMagModule(np.zeros(N), (0, 100.0 * nx, 0, 100.0 * ny, 0, 100.0 * nz), (100.0, 100.0, 100.0), (xp, yp, zp),
                         device=device, verbose=False)
    eng = gm._engine
    wm = gm.Wm.diagonal()
    d_true = eng.forward(wm * rho)
    dobs = d_true + np.random.default_rng(0).normal(0.0, 0.02 * np.abs(d_true).max(), N)
    eng.set_data(dobs)
    gm.dobs = dobs
    ones = np.ones(M)
    out = {"workload": workload, "chains": chains, "nsamples_per_chain": nsamples, "Lrange": [5, 20], "dt": dt}
    tmp = tempfile.mkdtemp(prefix="gravhmc_bench_")
    try:
        for sink in ("none", "binary"):
            st0 = eng.batch_resident_stats()
            with open(os.devnull, "w") as null, contextlib.redirect_stdout(null):
                t0 = time.perf_counter()
                acc_n, tot_n = g.HMCSampleBatch(gm, chains, nsamples, 0, dt, [5, 20], 0.001 * ones, 0.001 * ones,
                                                np.c_[0.0 * ones, ones], "mandatory", 1000, dobs, "Fixed", 0.8, 1.0,
                                                "Damping", 0.01, 100, 0.001, save_folder=os.path.join(tmp, sink + "_chain"),
                                                sample_sink=sink)
                eng.synchronize()
                el = time.perf_counter() - t0
            st1 = eng.batch_resident_stats()
            steps = st1["chain_steps"] - st0["chain_steps"]
            out[sink + "_sink"] = {"seconds": el, "trajectories": int(sum(tot_n)), "accepted": int(sum(acc_n)),
                                   "leapfrog_steps": int(steps), "leapfrog_steps_per_s": steps / el,
                                   "lock_step_launches": st1["launches"] - st0["launches"]}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    eng.close()
    return out


def c1_block(device, want_cpu):
    """BASELINE.json's target configuration (north_star: uniformgrid 20 x 30 x 10, Damping, one chain)
    measured in the same process, for the `extra` field of the line: steps/s, us per potential
    evaluation inside the resident chain kernel, and the CPU baseline at full size."""
    import gravinv3dhmc_amd as g
    name = "c1_uniform_20x30x10"
    mesh, xp, yp, zp, rho = make_problem(name)
    N, M = xp.size, mesh.size
    eng = g.Engine(N, M, device=device)
    eng.set_obs(xp, yp, zp)
    eng.set_cells(mesh.cell_bounds(), 0)
    eng.build_G()
    d_true = eng.forward(rho)
    wm = eng.weight(0.5)
    dobs = d_true + np.random.default_rng(0).normal(0.0, 0.02 * np.abs(d_true).max(), N)
    eng.set_data(dobs)
    eng.set_reg("Damping", 1.0, 0.01, mesh.shape, 0.001 * wm)
    eng.chain_init(0.001 * wm, 0.0 * wm, 1.0 * wm)
    steps, warmup, L, dt = 20000, 2000, 10, WORKLOADS[name][4]
    elapsed, nacc, ntraj, prof = run_single_chain(eng, M, 0.001, dt, L, steps, warmup, 100)
    cstat = eng.chain_stats()
    out = {"workload": name, "N_obs": int(N), "M_cells": int(M), "G_bytes": int(N) * int(M) * 8,
           "value": steps / elapsed, "unit": "leapfrog steps/s", "steps": steps, "warmup": warmup,
           "us_per_step": elapsed / steps * 1e6, "trajectories": ntraj, "accepted": nacc,
           "kernel": "resident_chain_kernel (G in LDS, %d launches)" % cstat["resident_launches"]
                     if cstat["resident_evaluations"] else "sweep_kernel per launch",
           "us_per_evaluation_in_kernel": prof["sweep_ms"] * 1e3 / max(1, prof["sweeps"]),
           "reference_formulation_equiv_GBps": 2 * N * M * 8 * steps / elapsed / 1e9}
    if cstat["resident_evaluations"]:
        out["roofline"] = resident_roofline(N, M, 1, prof, ntraj, cstat)
    if want_cpu:
        try:
            out["cpu_baseline"] = cpu_baseline(mesh, xp, yp, zp, dobs, target_s=6.0, max_cells=M)
        except Exception as e:
            out["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    eng.close()
    return out


#: aggregate LDS read bandwidth of the chip (MI355X_MICROARCH.md, LDS: "~150 TB/s for ds_read_b64/b128" with
#: every CU streaming) and the fp64 matrix peak (dense, 78.6 TFLOP/s)
LDS_PEAK_GBPS = 150000.0
FP64_MATRIX_PEAK_TFLOPS = 78.6


def resident_roofline(N, M, chains, prof, ntraj, cstat):
    """The roofline block of a run that stayed on the chip (G resident in LDS and registers: no HBM traffic
    to price).  One chain: both products of an evaluation read the resident operator once -- 2 N M 8 bytes
    out of LDS / registers -- against the chip's aggregate LDS bandwidth; the floor that actually binds is
    the exchange (three dependent hops between workgroups per evaluation, ~1 us each).  Several chains in
    lock-step (resident_batch_kernel): the products are 16-wide fp64 MFMA GEMMs, 4 N M 16 flop per lock-step
    of the batch, against the fp64 matrix peak."""
    evals = max(1, prof["sweeps"])
    us = prof["sweep_ms"] * 1e3 / evals
    lockstep = cstat.get("resident_batch_launches", 0) > 0
    if lockstep:
        flop = 4.0 * N * M * 16
        ach = flop / (us * 1e-6) / 1e12
        return {"bound": "mfma", "achieved": ach, "peak": FP64_MATRIX_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": ach / FP64_MATRIX_PEAK_TFLOPS, "traffic": None,
                "kernel": "resident_batch_kernel (%d chains in lock-step: G in LDS / registers, both products of all "
                          "chains as v_mfma_f64_16x16x4, one three-hop exchange of N x %d doubles per lock-step; %d "
                          "trajectories in %d launches)" % (chains, chains, ntraj, cstat["resident_batch_launches"]),
                "launches": cstat["resident_batch_launches"], "avg_ms": None, "lock_steps": prof["sweeps"],
                "us_per_lock_step": us, "flop_model": "4 N M 16 flop per lock-step (two 16-wide fp64 GEMMs; the MFMA "
                                                      "computes 16 chains whatever the batch holds)",
                "exchange_floor": "three dependent hops between workgroups per lock-step"}
    byts = 2.0 * N * M * 8
    ach = byts / (us * 1e-6) / 1e9
    return {"bound": "lds", "achieved": ach, "peak": LDS_PEAK_GBPS, "unit": "GB/s", "frac": ach / LDS_PEAK_GBPS,
            "traffic": None,
            "kernel": "resident_chain_kernel (%s%d trajectories in %d launches)"
                      % ("%d chains taking turns, " % chains if chains > 1 else "", ntraj, cstat["resident_launches"]),
            "launches": cstat["resident_launches"], "avg_ms": None, "evaluations": prof["sweeps"],
            "us_per_evaluation": us,
            "algorithmic_bytes_per_evaluation": byts,
            "byte_model": "2 N M 8: forward and adjoint each read the resident operator once (LDS; the adjoint's "
                          "copy sits in registers where it fits), priced against the aggregate LDS bandwidth",
            "exchange_floor": "three dependent hops between workgroups per evaluation (~3 of the us above): the "
                              "bound that binds; the LDS fraction says how far the local work is from mattering"}


def config_values(line):
    """Every BASELINE configuration's number in one flat map of the line itself (the driver's record keeps the
    top-level keys of the line but not the content of `extra`): tag -> [value in leapfrog (chain-)steps/s,
    roofline fraction of that run's dominant kernel]."""
    def pair(d):
        if not isinstance(d, dict) or "value" not in d:
            return None
        r = d.get("roofline") or {}
        f = r.get("frac", d.get("frac"))
        return [round(float(d["value"]), 1), None if f is None else round(float(f), 4)]
    out = {"c2": pair(line)}
    short = {"c1_uniform_20x30x10": "c1", "c1_uniform_16_chains": "c1_16", "c2_uniform_16_chains": "c2_16",
             "c2_uniform_16_chains_two_reads_of_G": "c2_16_2rd", "c3_segment_wavelet3d_tv": "c3",
             "c3_segment_wavelet3d_tv_16_chains": "c3_16",
             "c4_global_tesseroid_matrix_free": "c4_mf", "c4_global_tesseroid_dense": "c4_dense",
             "c4_global_tesseroid_shift_invariant": "c4_si",
             "c4_global_tesseroid_shift_invariant_8_chains": "c4_si8",
             "c4_global_tesseroid_matrix_free_8_chains": "c4_mf8", "c5_share_of_one_gpu_of_8": "c5_share",
             "x3_global_one_degree_shift_invariant": "g1deg_si",
             "x3_global_one_degree_shift_invariant_8_chains": "g1deg_si8",
             "c5_share_row_blocks": "c5_rows"}
    for tag, d in (line.get("extra") or {}).items():
        if tag in short:
            out[short[tag]] = pair(d)
    hb = (line.get("extra") or {}).get("c1_hmcsample_batch_16_chains")
    if isinstance(hb, dict) and "none_sink" in hb:
        out["c1_16_sampler"] = [round(hb["none_sink"]["leapfrog_steps_per_s"], 1), None]
    h4 = (line.get("extra") or {}).get("c4_hmcsample")
    if isinstance(h4, dict) and "binary_sink" in h4:
        out["c4_sampler"] = [round(h4["binary_sink"]["leapfrog_steps_per_s"], 1), None]
    hs = (line.get("extra") or {}).get("c2_hmcsample")
    if isinstance(hs, dict) and "binary_sink" in hs:
        out["c2_sampler"] = [round(hs["binary_sink"]["leapfrog_steps_per_s"], 1),
                             round(hs.get("sampler_vs_run_chain", 0.0), 3)]
    return out


def launch_ranks(n, argv):
    """One process per GPU through torch.distributed.run on 127.0.0.1 (a free port), the command the
    driver itself uses for N > 1; stdout / stderr pass through, the return code is the launcher's."""
    import socket
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    rc = 1
    for attempt in range(2):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
        t0 = time.time()
        rc = subprocess.run(cmd, env=env).returncode
        # Between close() and the launcher's bind another process may take the port: a launch that dies
        # within seconds (the rendezvous could not bind) is tried once more on a fresh port.
        if rc == 0 or time.time() - t0 > 20.0:
            break
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--workload", default="c2_uniform_100x100x50",
                    choices=list(WORKLOADS) + list(EXTRA))
    ap.add_argument("--batch-team", choices=["auto", "on", "off"], default="auto",
                    help="several chains per GPU on the stored kernel: teams of workgroups that read G once per "
                         "leapfrog step (csrc/batchteam.hip.h) / two reads with a second copy of G / the library's choice")
    ap.add_argument("--chains-per-gpu", type=int, default=1,
                    help="independent chains batched on each GPU against ONE copy of G through the "
                         "fp64 MFMA path (1..16); value then counts the steps of all chains")
    ap.add_argument("--cells-fraction", type=int, default=1,
                    help="keep only the first 1/F of the cells: the share one of F GPUs holds when "
                         "the chain is sharded (single-GPU rehearsal of a model that exceeds one HBM)")
    ap.add_argument("--rows-fraction", type=int, default=1,
                    help="keep only the first 1/F of the observations: the share one of F GPUs holds when the chain is "
                         "sharded in ROW blocks (--shard --shard-axis rows; single-GPU rehearsal)")
    ap.add_argument("--matrix-free", action="store_true",
                    help="never store G: re-evaluate the kernel entries in every pass")
    ap.add_argument("--shift-invariant", action="store_true",
                    help="regular spherical grids (C4): keep the longitude-shift-invariant table instead of G "
                         "(gh_set_shift_invariant: 35 MB instead of 4.25 GB)")
    ap.add_argument("--traj-len", type=int, default=10, help="leapfrog steps per trajectory")
    ap.add_argument("--hmcsample", type=int, default=0, metavar="NSAMPLES",
                    help="measure the SAMPLER (HMCSample, binary and text sinks, Lrange [5,20]) on the workload "
                         "until NSAMPLES proposals are accepted, print its block and exit")
    ap.add_argument("--hmcsample-batch", type=int, default=0, metavar="NSAMPLES",
                    help="measure the BATCH sampler (HMCSampleBatch, --chains-per-gpu chains, sinks none and binary) on the "
                         "workload until every chain has NSAMPLES accepted samples, print its block and exit")
    ap.add_argument("--seed", type=int, default=100,
                    help="np.random.seed of the chain of rank 0; rank r takes seed + r (hmc.py:369)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="skip the extra block (C1, north_star's target configuration) of the default line")
    ap.add_argument("--shard", action="store_true",
                    help="ONE chain whose cells (columns of G) are split over the ranks' GPUs, "
                         "all-reduce of the forward partial per step (strong scaling); default "
                         "is one independent chain per GPU (weak scaling)")
    ap.add_argument("--shard-backend", default="rccl", choices=["rccl", "gloo"])
    ap.add_argument("--shard-axis", default="cells", choices=["cells", "rows"],
                    help="--shard: what is split over the ranks -- the cells (column blocks of G: all-reduce of N + 2 doubles "
                         "per step, ONE read of the shard) or the observations (row blocks, BASELINE configs[4] as worded: "
                         "all-reduce of the M-vector gradient + two scalars per step, TWO reads of the shard)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="all ranks share GPU 0 (rehearsal of the N>1 launch path on a 1-GPU box)")
    args = ap.parse_args()
    if args.batch_team != "auto":
        os.environ["GRAVHMC_BATCH_TEAM"] = "1" if args.batch_team == "on" else "0"

    if args.hmcsample > 0 and args.workload == "c4_global_tesseroid":
        print(json.dumps({"hmcsample": hmcsample_global_block(int(os.environ.get("LOCAL_RANK", "0")), args.hmcsample)}))
        return
    if args.hmcsample > 0:
        print(json.dumps({"hmcsample": hmcsample_block(int(os.environ.get("LOCAL_RANK", "0")), args.workload,
                                                       args.hmcsample)}))
        return
    if args.hmcsample_batch > 0:
        print(json.dumps({"hmcsample": hmcsample_batch_block(int(os.environ.get("LOCAL_RANK", "0")), args.workload,
                                                             max(2, args.chains_per_gpu), args.hmcsample_batch)}))
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` as the driver types it for N = 1: start the N ranks ourselves, as
        # CHILD processes of a parent that has not touched the GPU (no HIP call, no library load so
        # far; never an exec of a process that has), relay rank 0's JSON line, exit with their status.
        # The reference's counterpart is `mpiexec -n K python main_*.py` (example/uniformgrid/run_main.sh:17).
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    # control plane only (barrier + max of the elapsed time): the chains are independent, so
    # no tensor of the data path ever crosses ranks.  gloo on CPU tensors keeps torch's own HIP
    # runtime out of the process (gravinv3dhmc_amd/dist.py, covered by tests/test_dist_gloo.py).
    from gravinv3dhmc_amd.dist import Ranks
    ranks = Ranks()
    rank, local_rank, world = ranks.rank, ranks.local_rank, ranks.world
    if world != args.gpus:
        sys.exit("bench.py --gpus %d runs under a launcher with WORLD_SIZE = %d" % (args.gpus, world))

    import gravinv3dhmc_amd as g
    extra = EXTRA.get(args.workload)
    if extra:
        mesh, (xp, yp, zp), rho = make_extra(args.workload)
    else:
        mesh, xp, yp, zp, rho = make_problem(args.workload)
        extra = dict(kind=0, reg="Damping", alpha=1.0, beta=0.01, dt=WORKLOADS[args.workload][4],
                     hi=1.0, wavelet=0)
    bounds = mesh.cell_bounds()
    if args.cells_fraction > 1:
        keep = mesh.size // args.cells_fraction
        bounds, rho = bounds[:keep], rho[:keep]
    if args.rows_fraction > 1:
        keep = xp.size // args.rows_fraction
        xp, yp, zp = xp[:keep], yp[:keep], zp[:keep]
    N, M = xp.size, bounds.shape[0]
    dev = 0 if args.rehearse_on_one_gpu else local_rank
    if args.shard:
        from gravinv3dhmc_amd.dist import make_sharded_engine
        eng = make_sharded_engine(N, M, ranks, device=dev, backend=args.shard_backend, axis=args.shard_axis)
    else:
        eng = g.Engine(N, M, device=dev)
    info = eng.device_info()
    if args.matrix_free:
        eng.set_matrix_free(True)
    if args.shift_invariant:
        eng.set_shift_invariant(True)
    t0 = time.time()
    eng.set_obs(xp, yp, zp)
    eng.set_cells(bounds, extra["kind"], 1.6)
    eng.build_G()
    eng.synchronize()
    t_build = time.time() - t0
    d_true = eng.forward(rho)                          # noise-free synthetic data, unweighted G
    t0 = time.time()
    wm = eng.weight(0.5)
    t_weight = time.time() - t0
    nnz = None
    if extra["wavelet"]:
        nnz, _ = eng.compress_wavelet(extra["wavelet"], mesh.shape, 1e-3, 2)
    dobs = d_true + np.random.default_rng(0).normal(0.0, 0.02 * np.abs(d_true).max(), N)
    eng.set_data(dobs)
    eng.set_reg(extra["reg"], extra["alpha"], extra["beta"], mesh.shape, 0.001 * wm)
    low, high = 0.0 * wm, extra["hi"] * wm
    eng.chain_init(0.001 * wm, low, high)

    Sigma, dt, L = 0.001, extra["dt"], args.traj_len
    CPG = args.chains_per_gpu
    barrier = ranks.barrier
    if CPG > 1:
        # ---- several chains per GPU: one RandomState per chain (seed 100 + global chain index,
        # the stream np.random.seed gives the reference's rank), lock-step rounds of L steps
        import threading
        from concurrent.futures import ThreadPoolExecutor
        from gravinv3dhmc_amd.inversion.rng import LegacyDraws, draw_workers
        pool = ThreadPoolExecutor(max_workers=2)
        # (a draw is mostly sequential: one generator per core, no helpers -- rng.draw_workers)
        n_draw = draw_workers(CPG)
        draw_pool = ThreadPoolExecutor(max_workers=n_draw)
        x0s = np.stack([0.001 * wm for _ in range(CPG)])
        eng.batch_init(x0s, low, high)
        # (trajectories offered per chain and call: the sampler's rule, inversion/hmc.py HMCSampleBatch)
        Tmax = int(max(2, min(32, (256 << 20) // (8 * M * CPG))))

        class Rounds(object):
            """Every chain runs total_steps leapfrog steps in trajectories of L (the last one shorter), through
            the sampler's path (HMCSampleBatch): gh_batch_run in carry-over mode, up to Tmax (<= 32) trajectories per
            chain offered per call -- a finishing chain's last step takes the first step of the next one it
            has been offered -- the following offers drawn meanwhile, each chain from its own legacy stream
            (bit for bit np.random.RandomState(seed + chain): randn(M) * Sigma, rand() per trajectory) by the
            library's generator, one thread per chain."""

            def __init__(self, total_steps, seed0):
                self.plan = [L] * (total_steps // L) + ([total_steps % L] if total_steps % L else [])
                # (each chain draws into a ring of page-locked rows: the library sends them to the GPU from there)
                self.draws = [LegacyDraws(M, (L, L), Sigma, fixed_L=self.plan, seed=seed0 + k,
                                          helpers=1 if n_draw >= 8 else None).use_ring(eng, 2 * Tmax)
                              for k in range(CPG)]
                self.queue = [[] for _ in range(CPG)]      # per chain: (n, p0, u) drawn, not started yet
                self.top_up()                              # a sampler in steady state has its next offer drawn

            def top_up(self):
                def one(k):
                    n = 2 * Tmax - len(self.queue[k])
                    if n > 0:
                        self.queue[k].extend(self.draws[k].take_ring(n))   # (L, row address, u, ring index)
                list(draw_pool.map(one, range(CPG)))

            def run(self):
                queue, nacc, done = self.queue, 0, [0] * CPG
                clk = HOST_LOOP
                while min(done) < len(self.plan):
                    t_a = time.perf_counter()
                    T = min(Tmax, min(len(q) for q in queue))
                    if T == 0:                             # nothing left to offer: finish what is in flight
                        acc, _, _, ns, nd = eng.batch_run([[] for _ in range(CPG)], dt, np.zeros((CPG, 0)),
                                                          np.zeros((CPG, 0)), carry=True)
                    else:
                        entered = threading.Event()
                        fut = pool.submit(eng.batch_run, [[tr[1] for tr in q[:T]] for q in queue], dt,
                                          [[tr[0] for tr in q[:T]] for q in queue],
                                          [[tr[2] for tr in q[:T]] for q in queue], False, True, entered)
                        # (the GPU has its work before a dozen drawing threads compete for the interpreter lock)
                        while not entered.wait(0.05):
                            if fut.done():
                                break
                        self.top_up()                      # (the offers after this one, while the GPU runs)
                        t_b = time.perf_counter()
                        acc, _, _, ns, nd = fut.result()
                        clk["draw_s"] += t_b - t_a
                        clk["wait_for_gpu_call_s"] += time.perf_counter() - t_b
                        clk["calls"] += 1
                    for k in range(CPG):
                        del queue[k][:int(ns[k])]
                        done[k] += int(nd[k])
                        nacc += int(acc[k, :int(nd[k])].sum())
                return nacc, len(self.plan)

            def release(self):
                # (outside the timed region: un-pinning the rings -- 512 MB at C2 with 16 chains -- takes longer than a step)
                for d in self.draws:
                    d.release()

        HOST_LOOP = {"draw_s": 0.0, "wait_for_gpu_call_s": 0.0, "calls": 0}
        if args.warmup > 0:
            wr = Rounds(args.warmup, args.seed + 1000 + rank * CPG)
            wr.run()
            wr.release()
        rounds = Rounds(args.steps, args.seed + rank * CPG)
        HOST_LOOP.update(draw_s=0.0, wait_for_gpu_call_s=0.0, calls=0)
        eng.synchronize()
        barrier()
        eng.profile_enable(True)
        t0 = time.perf_counter()
        naccept, ntraj = rounds.run()
        eng.synchronize()
        elapsed = time.perf_counter() - t0
        rounds.release()
        barrier()
    else:
        # the reference's RNG stream (legacy global generator), one chain per rank (a sharded chain
        # shares one stream)
        elapsed, naccept, ntraj, prof = run_single_chain(eng, M, Sigma, dt, L, args.steps, args.warmup,
                                                         args.seed if args.shard else args.seed + rank, barrier)
    if CPG > 1:
        prof = eng.profile_read()
        eng.profile_enable(False)
    # what a pure read of the same matrix reaches on this device (outside the timed region)
    stream_gbps = None
    if not args.matrix_free and not args.shift_invariant and int(N) * int(M) * 8 >= (1 << 30):
        stream_gbps = eng.stream_read_gbps(nt=True, reps=3)
    elapsed = ranks.max(elapsed)
    final_U_ranks = ranks.gather(LAST_STATE["U"])      # rank 0: every rank's chain state, in rank order

    if rank == 0:
        traffic, traffic_src = pmc_traffic(args.workload)
        sweep_ms = prof["sweep_ms"] / max(1, prof["sweeps"])
        bytes_sweep = prof["bytes_per_sweep"]          # N*M_local*8: one read of this rank's G
        achieved = bytes_sweep / (sweep_ms * 1e-3) / 1e9
        line = {
            "metric": "HMC leapfrog steps/sec + G\u00b7\u03c1 achieved HBM GB/s at 1/2/4/8 GPUs",
            "value": args.steps * CPG * (1 if args.shard else world) / elapsed,
            "unit": "leapfrog steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if args.shard else "weak",
            "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": args.workload, "N_obs": int(N), "M_cells": int(M),
                       "G_bytes": int(N) * int(M) * 8, "regulariser": extra["reg"],
                       "matrix_free": bool(args.matrix_free), "shift_invariant": bool(args.shift_invariant),
                       "wavelet_nnz": nnz,
                       "chains_per_gpu": CPG, "dt": dt, "traj_len": L, "trajectories": ntraj,
                       # several chains: the host loop's clocks (drawing the next offers beside the GPU call, then
                       # waiting for the call)
                       "host_loop": ({k: (round(v, 4) if isinstance(v, float) else v) for k, v in HOST_LOOP.items()}
                                     if CPG > 1 else None),
                       "accepted": naccept, "final_U": LAST_STATE["U"] if CPG == 1 else None,
                       "final_U_per_rank": final_U_ranks if CPG == 1 and world > 1 else None,
                       "seed": args.seed,
                       "speculative_first_steps": eng.chain_stats() if CPG == 1 else None, "parallelism": (("1 chain, observations (row blocks) sharded x%d, %s all-reduce of the M-vector gradient + 2 "
                                        "scalars per step, two reads of the shard per step"
                                        if args.shard_axis == "rows" else
                                        "1 chain, cells sharded x%d, %s all-reduce of N+2 doubles per step")
                                       % (world, args.shard_backend)) if args.shard
                       else "chain-parallel x%d (no collective)" % world,
                       "device": info["name"], "cus": info["cus"],
                       "G_build_s": round(t_build, 3), "weighting_s": round(t_weight, 3)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic, "traffic_from_profile": traffic_src,
                         "kernel": "sweep_kernel (fused adjoint+update+forward, one read of G)",
                         "launches": prof["sweeps"], "avg_ms": sweep_ms,
                         "algorithmic_bytes_per_launch": bytes_sweep,
                         "reference_formulation_equiv_GBps":
                             2 * bytes_sweep * args.steps / elapsed / 1e9,
                         # a plain read-only pass over the same matrix on this device (SURVEY 8d's
                         # "attainable" microbenchmark; the sweep's access pattern reaches or beats it)
                         "stream_read_microbench_GBps": stream_gbps,
                         "vs_stream_read_microbench": achieved / stream_gbps if stream_gbps else None},
        }
        cstat = eng.chain_stats()
        if args.shard and args.shard_axis == "rows":
            line["roofline"]["kernel"] = ("sweep_kernel, adjoint-only and forward-only passes over the rank's row block (the "
                                          "gradient is all-reduced between them: two reads of the shard per leapfrog step)")
        if cstat.get("team_launches", 0) > 0:
            # N > 16384: the timed launches were team sweeps (csrc/teamsweep.hip.h)
            line["roofline"]["kernel"] = ("teamsweep_kernel (teams of %d workgroups share each column: fused "
                                          "adjoint+update+forward, one read of G)" % cstat["team_members"])
        if cstat.get("resident_evaluations", 0) > 0:
            # G never left the chip: the figure below is what the reference formulation would have
            # had to read per second, not HBM traffic
            line["roofline"].update(resident_roofline(N, M, CPG, prof, ntraj, cstat))
        if CPG > 1 and cstat.get("resident_evaluations", 0) > 0:
            line["roofline"]["reference_formulation_equiv_GBps"] = \
                2 * bytes_sweep * CPG * args.steps / elapsed / 1e9
        elif CPG > 1:
            line["roofline"].update({
                "kernel": "batch_adjoint_kernel + batch_forward_kernel (v_mfma_f64_16x16x4, %d chains "
                          "share each read of G; two sweeps per leapfrog step of the batch)" % CPG,
                "fp64_matrix_TFLOPs": 4.0 * N * M * 16 * prof["sweeps"] / 2 / (prof["sweep_ms"] * 1e-3) / 1e12,
                "reference_formulation_equiv_GBps": 2 * bytes_sweep * CPG * args.steps / elapsed / 1e9})
        if CPG > 1 and not args.matrix_free and cstat.get("resident_evaluations", 0) == 0:
            tstat = eng.batch_fused_stats()
            if tstat["launches"] > 0 and tstat["timeouts"] == 0:
                # the timed launches were team passes (csrc/batchteam.hip.h): both products of all chains
                # from ONE read of G -- a launch is a leapfrog step of the batch
                line["roofline"].update({
                    "kernel": ("batch_team_kernel (%d chains share ONE read of G per leapfrog step: teams of %d "
                               "workgroups x %d ranges of column tiles, v_mfma_f64_16x16x4 for both products)"
                               % (CPG, tstat["members"], tstat["ranges"])),
                    "fp64_matrix_TFLOPs": 4.0 * N * M * 16 * prof["sweeps"] / (prof["sweep_ms"] * 1e-3) / 1e12,
                    "fp64_matrix_peak_TFLOPs": 78.6,
                    "reference_formulation_equiv_GBps": 2 * bytes_sweep * CPG * args.steps / elapsed / 1e9})
        if args.shift_invariant:
            # K[i, (c, k)] = T[c][class_i][(m_i - k) mod n]: the pass reads the table once per step
            si = eng.shift_invariant_info()
            hm = eng.shift_invariant_harmonic()
            secs = prof["sweep_ms"] * 1e-3
            if hm["on"]:
                # harmonic domain: n_rows x n_classes x n_freq complex multiply-adds per product; what the pass
                # must move is the complex table T^ (beyond L2: Infinity Cache / HBM), once per step
                streamed = hm.get("form") == "streamed"
                # (the streamed form for large grids, csrc/lonsymw.hip.h, reads T^ twice: row-parallel adjoint, forward product)
                byts = float(hm["table_bytes"]) * (2.0 if streamed else 1.0)
                ach = byts * prof["sweeps"] / secs / 1e9 if secs > 0 else None
                line["roofline"] = {
                    "bound": "hbm", "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                    "frac": ach / 8000.0 if ach else None, "traffic": None,
                    "kernel": ("lonsymw_rhat_kernel + lonsymw_sweep_kernel + lonsymw_forward_kernel + lonsymw_post_kernel "
                               "(longitude-harmonic domain, STREAMED: %d frequencies x %d classes x %d cell rows, complex "
                               "table T^ of %.1f MB read twice per step -- adjoint per cell row, forward product per "
                               "(class, frequency) over ranges of rows; transforms of length %d; the dense kernel of this "
                               "grid would be %.0f GB)"
                               % (hm["n_freq"], si["n_classes"], si["n_rows"], hm["table_bytes"] / 1e6, si["n_lon"],
                                  N * M * 8 / 1e9)) if streamed else
                              "lonsymh_rhat_kernel + lonsymh_sweep_kernel + lonsymh_post_kernel (longitude-harmonic "
                              "domain: %d frequencies x %d classes x %d cell rows, complex table T^ of %.1f MB read once "
                              "per step by %d workgroups; transforms of length %d inside the cell row's workgroup)"
                              % (hm["n_freq"], si["n_classes"], si["n_rows"], byts / 1e6, hm["workgroups"], si["n_lon"]),
                    "launches": prof["sweeps"], "avg_ms": sweep_ms, "table": dict(si, harmonic=hm),
                    "algorithmic_bytes_per_launch": byts,
                    "byte_model": "the complex table T^[row][class][frequency] (16 bytes per entry), read %s per pass"
                                  % ("twice" if streamed else "once"),
                    "note": ("four plain launches per pass, no inter-workgroup waits" if streamed else "three dependent, latency-bound launches per pass") + "  (time between the first one's start and "
                            "the last one's end)" + ("; %d chains on their own streams: the passes overlap, their "
                                                     "durations are summed" % CPG if CPG > 1 else ""),
                    "dense_G_equiv_GBps": N * M * 8 / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else None}
                rst = eng.shift_invariant_resident_stats()
                if CPG == 1 and rst["launches"] > 0:
                    # ONE persistent launch per batch of trajectories (csrc/lonres.hip.h): the table never leaves the
                    # workgroups' registers; what moves per evaluation is the exchange -- every workgroup's D^ partial
                    # out, the clusters' sums, R^ back to every workgroup -- through the memory system ("all bytes
                    # leave L2"), in four dependent hops
                    E16 = 16.0 * si["n_classes"] * hm["n_freq"]
                    xbytes = rst["workgroups"] * E16 * 2 + 8 * 2 * E16 * 2
                    ach = xbytes * prof["sweeps"] / secs / 1e9 if secs > 0 else None
                    line["roofline"].update({
                        "achieved": ach, "frac": ach / 8000.0 if ach else None,
                        "kernel": "lonsymh_resident_kernel (ONE persistent launch per batch of trajectories: %d workgroups "
                                  "keep their cell rows' part of the complex table T^ (%.1f MB) in REGISTERS; per evaluation "
                                  "forward partials -> cluster sums -> class owners (inverse transform, residuals, forward "
                                  "transform) -> R^ to every workgroup -> adjoint, gradient, leapfrog update; Metropolis "
                                  "test inside)" % (rst["workgroups"], byts / 1e6),
                        "algorithmic_bytes_per_launch": xbytes,
                        "byte_model": "exchange of one evaluation: %d workgroups x %d complex entries (16 B) out and R^ back, "
                                      "8 cluster sums of tagged granules (32 B per entry); the table itself is not read again"
                                      % (rst["workgroups"], si["n_classes"] * hm["n_freq"]),
                        "note": "launches = evaluations inside the persistent launches (forward + adjoint product each), avg_ms "
                                "= kernel time per evaluation; the floor is the latency of the dependent exchange hops "
                                "(DESIGN 4.9), not bandwidth",
                        "persistent": rst})
            else:
                flops = 4.0 * N * M * prof["sweeps"]
                line["roofline"] = {
                    "bound": "fp64 vector out of LDS (shift-invariant table, %.1f MB, cache resident; the N*M "
                             "multiply-adds of adjoint and forward remain)" % (si["table_bytes"] / 1e6),
                    "achieved": flops / secs / 1e12 if secs > 0 else None, "peak": FP64_VECTOR_PEAK_TFLOPS,
                    "unit": "TFLOP/s", "frac": flops / secs / 1e12 / FP64_VECTOR_PEAK_TFLOPS if secs > 0 else None,
                    "traffic": None,
                    "kernel": "lonsym_sweep_kernel (workgroup = cell row of %d longitudes: %d observation classes in "
                              "the lanes, 8-shift register window, fused adjoint+update+forward)"
                              % (si["n_lon"], si["n_classes"]),
                    "launches": prof["sweeps"], "avg_ms": sweep_ms, "table": si,
                    "table_read_GBps": si["table_bytes"] / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else None,
                    "flop_model": "2 N M multiply-adds per pass (adjoint + forward) = 4 N M flop",
                    "dense_G_equiv_GBps": N * M * 8 / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else None}
        elif args.matrix_free:
            # no stored G: the pass is bound by fp64 vector arithmetic.  FLOP model: the
            # operations of the reference's formulas as written, every +, -, *, /, sqrt and
            # trigonometric / logarithm call counted as ONE flop, times the work the kernel counted
            # itself (GLQ leaves of the adaptive tesseroid engine / prism entries).
            st = eng.matrix_free_stats()
            fus = eng.batch_fused_stats() if CPG > 1 else {"launches": 0, "members": 0, "ranges": 0, "timeouts": 0}
            teams = fus["launches"] > 0 and fus["timeouts"] == 0
            tms = eng.matrix_free_team_stats() if CPG == 1 else {"launches": 0, "members": 0, "ranges": 0}
            tess = extra["kind"] == 1
            near = tess and st["near_entries"] > 0
            # executed per entry: prisms the whole entry; tesseroids the root leaf (with the near-field
            # table: the leaf's pair-dependent part only -- what depends on the cell alone and the
            # distance/size test are evaluated once at build time) or the adaptive engine's leaves
            per_unit = (FLOPS_PER_TESS_PAIR_HOISTED if near else FLOPS_PER_TESS_LEAF) if tess \
                else FLOPS_PER_PRISM_ENTRY
            timed_launches = max(1, prof["sweeps"])
            share = timed_launches / max(1, st["launches"])   # (events cover at most 4096 launches)
            units = (st["entries"] if near else st["leaves"]) * share
            secs = prof["sweep_ms"] * 1e-3
            tflops = units * per_unit / secs / 1e12 if secs > 0 else None
            # the same step priced as the reference formulates it: every pair through the adaptive
            # engine (distance/size test + leaf, 244 flop per GLQ leaf)
            ref_leaves = (st["entries"] - st["near_entries"] * st["launches"] + st["near_leaves"] * st["launches"]) \
                if near else st["leaves"]
            ref_unit = FLOPS_PER_TESS_LEAF if tess else FLOPS_PER_PRISM_ENTRY
            line["roofline"] = {
                "bound": "fp64 vector (no stored G: entries re-evaluated, every entry %s per step)"
                         % (("once for all chains together" if teams else "twice for all chains together")
                            if CPG > 1 else "once"),
                "achieved": tflops, "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": tflops / FP64_VECTOR_PEAK_TFLOPS if tflops else None, "traffic": None,
                "kernel": (("mfb_fused_kernel (%d chains share every evaluated entry: teams of %d workgroups x %d "
                            "ranges of column tiles, 16 columns x 448 rows staged in LDS, partial dots exchanged "
                            "through memory, v_mfma_f64_16x16x4 for the chains; ONE evaluation per entry and step "
                            "of the whole batch)" % (CPG, fus["members"], fus["ranges"])) if teams else
                           ("mfb_adjoint_kernel + mfb_forward_kernel (%d chains share every evaluated entry: "
                            "16 columns x 512 rows staged in LDS, v_mfma_f64_16x16x4 for the chains; two "
                            "evaluations per entry and step of the whole batch)" % CPG)) if CPG > 1 else
                          ("mf_team_kernel (teams of %d workgroups x %d ranges of column tiles: a wave keeps its "
                           "column's dot as one number per tile, 16 doubles per member and tile cross the team; "
                           "every entry evaluated once)" % (tms["members"], tms["ranges"])) if tms["launches"] > 0 else
                          "mf_tess_fast_kernel / mf_fused_kernel (entries of a cell's column evaluated once, "
                          "dot with r, leapfrog update, forward accumulation)",
                "launches": st["launches"], "avg_ms": sweep_ms, "team_form": fus if CPG > 1 else None,
                "entries_per_launch": st["entries"] / max(1, st["launches"]),
                "leaves_per_launch": st["leaves"] / max(1, st["launches"]),
                "near_field_table": {"entries": st["near_entries"], "glq_leaves": st["near_leaves"],
                                     "bytes": st["near_entries"] * 12} if tess else None,
                "flop_model": "%d flop per %s (formulas as written; +, -, *, /, sqrt, cos, log, atan2 = 1 flop each)"
                              % (per_unit, ("entry: pair-dependent part of the root GLQ leaf" if near else
                                            "GLQ leaf incl. its distance/size test") if tess else "prism entry"),
                "entries_per_s": st["entries"] * share / secs if secs > 0 else None,
                "reference_formulation_equiv_TFLOPs": ref_leaves * share * ref_unit / secs / 1e12 if secs > 0 else None}
        if not args.no_cpu_baseline and world == 1 and args.workload in WORKLOADS:
            try:
                line["cpu_baseline"] = cpu_baseline(mesh, xp, yp, zp, dobs)
            except Exception as e:  # the baseline is a reported extra, never the product path
                line["cpu_baseline"] = {"error": "%s: %s" % (type(e).__name__, e)}
    eng.close()
    if rank == 0:
        if world == 1 and args.workload == "c2_uniform_100x100x50" and CPG == 1 and not args.shard \
                and not args.matrix_free and not args.no_extra:
            # north_star's target configuration next to the headline workload, same process
            try:
                line["extra"] = {"c1_uniform_20x30x10": c1_block(dev, not args.no_cpu_baseline)}
            except Exception as e:
                line["extra"] = {"c1_uniform_20x30x10": {"error": "%s: %s" % (type(e).__name__, e)}}
            for tag, xargs in EXTRA_RUNS:
                try:
                    line["extra"][tag] = extra_run(dev, xargs)
                except Exception as e:
                    line["extra"][tag] = {"error": "%s: %s" % (type(e).__name__, e)}
            line["config_values"] = config_values(line)
        print(json.dumps(line))
    ranks.close()


if __name__ == "__main__":
    main()
