/*
 * gravhmc.h -- C-ABI of the MI355X-native HMC gravity-inversion hot path (libgravhmc.so).
 *
 * The reference (ChuWeiEr/GravInv3DHMC) is pure Python with no FFI/plugin registry; the seam
 * its sampler uses is the duck-typed model object of inversion/hmc.py:30-32,71-83 (calls
 * `model.kernelw()` and `model.misfit_and_grad(...)`) and, one level down, the native kernels
 * `gravmag/_prism.pyx:265-290` (`_prism.gz`) and `gravmag/_tesseroid_numba.py:32-71,335`
 * (`_tesseroid_numba.gz`).  Every entry point below names the reference interface it replaces.
 * INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C: opaque context, host pointers to caller-owned contiguous float64 buffers, sizes
 *     as int64_t.  No Python / torch types.  Calls are synchronous on return.
 *   - every function returns 0 on success or a negative gh_status; the text of the last error
 *     is available from gh_last_error(ctx) (or gh_last_error(NULL) for gh_create failures).
 *   - a context is bound to one GPU and one HIP stream and is NOT thread-safe (the reference is
 *     single-threaded per chain; chains are separate processes, hmc.py:367-369).
 *   - dense matrices are column-major N x M ("Fortran order": one cell = one contiguous column
 *     of N observations), the layout the reference ends up with for Aw (potential.py:259).
 *   - there is no CPU fallback: without a usable HIP device gh_create fails.
 *
 * Environment switches read by the library (diagnostics and A/B measurements; the defaults are what
 * the tests pin and what DESIGN.md's numbers were measured with).  Part of this interface:
 *   code path, SAME arithmetic to rounding of the summation order (results agree to <= 1e-12):
 *     GRAVHMC_RESIDENT=0 (no resident chain kernel), GRAVHMC_RESIDENT_LOCAL / _REGS / _STREAM /
 *     _STREAM_MB / _CT / _WGS (its variants), GRAVHMC_TEAM=0 / GRAVHMC_TEAM_LAG (N > 16384: row panels
 *     / lag of the team sweep), GRAVHMC_EPILOGUE1=0 (two-launch epilogue), GRAVHMC_MF_FUSED=0 /
 *     GRAVHMC_MF_PIPE=0 / GRAVHMC_MF_NEAR=0 (matrix-free: two-pass form / plain build / subdivision
 *     inside the pass instead of the near-field table), GRAVHMC_MFB_FUSED=0 (matrix-free batch: two
 *     passes instead of teams), GRAVHMC_MFB_RU=0 (no one-height specialisation), GRAVHMC_BATCH_SPEC=0,
 *     GRAVHMC_BATCH_RELAYOUT=0, GRAVHMC_BATCH_TEAM=0 (stored-kernel batch: two reads of G per step with a second
 *     copy of G instead of teams reading it once), GRAVHMC_RESIDENT_BATCH=0 (chains of a batch take turns in the
 *     resident kernel instead of running in lock-step), GRAVHMC_LONSYM_HARMONIC=0 / GRAVHMC_LONSYM_FUSED=1 /
 *     GRAVHMC_LONSYM_RESIDENT=0 (shift-invariant store: direct correlations / one-launch epilogue / one launch per
 *     phase instead of the persistent launch), GRAVHMC_LONSYM_WIDE=0 / 2 (the streamed harmonic form of large grids:
 *     off / also where the register form applies), GRAVHMC_LONSYM_W, GRAVHMC_LW_WAVES_PER_CU, GRAVHMC_LW_FWD, GRAVHMC_LW_MIRROR=0 (one row of T^ per cell row),
 *     GRAVHMC_DWT_LDS / _MAX (one-launch wavelet transform);
 *   arithmetic of an entry (within the path's stated 1e-10, ~1e-14 measured): GRAVHMC_MF_EXACT -- the
 *     DEFAULT of gh_set_matrix_free_exact only; that call overrides it;
 *   diagnostics that BREAK the results (timing only): GRAVHMC_LW_BREAK (phases of lonsymw_sweep_kernel off),
 *     GRAVHMC_BT_BREAK; GRAVHMC_LW_LDS_PAD (fewer workgroups per CU, results intact);
 *   tuning without any effect on results: GRAVHMC_PF, _NT, _TW, _TW8, _WG_PER_CU, _MIN_COLS,
 *     _INFLIGHT_MB, GRAVHMC_MF_T, _MF_WG_PER_CU, GRAVHMC_MFB_WG_PER_CU, _MFB_RANGES, GRAVHMC_RNG_THREADS;
 *   test hooks (force a time-out path): GRAVHMC_TEAM_TEST_ABORT, GRAVHMC_RESIDENT_TEST_ABORT,
 *     GRAVHMC_MFB_TEST_ABORT, GRAVHMC_BATCH_TEAM_TEST_ABORT, GRAVHMC_MF_TEAM_TEST_ABORT, GRAVHMC_RESBATCH_TEST_ABORT,
 *     GRAVHMC_LONRES_TEST_ABORT; timing experiments that BREAK results: GRAVHMC_MFB_DBG, GRAVHMC_BT_BREAK, and
 *     GRAVHMC_RESIDENT_TIMING, GRAVHMC_MFB_TIMING (+ GRAVHMC_BT_DBG_MEM / _WAVE: whose), GRAVHMC_LONSYM_TIMING
 *     (per-phase clocks, results intact).
 * (Python side: GRAVHMC_HOST_RNG=numpy draws with np.random itself -- same stream; GRAVHMC_LIB = path
 * of the shared library.)
 */
#ifndef GRAVHMC_H
#define GRAVHMC_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gh_ctx gh_ctx;

typedef enum {
    GH_OK = 0,
    GH_ERR_ARG = -1,      /* bad argument / call order (ValueError in the reference) */
    GH_ERR_HIP = -2,      /* HIP runtime error, text in gh_last_error */
    GH_ERR_NOMEM = -3,    /* device allocation failed */
    GH_ERR_OVERFLOW = -4, /* tesseroid subdivision stack > 100 entries (OverflowError,
                             _tesseroid_numba.py:53-54) */
    GH_ERR_UNSUPPORTED = -5,
    GH_ERR_COMM = -6      /* collective layer error */
} gh_status;

enum { GH_CELL_PRISM = 0, GH_CELL_TESSEROID = 1 };
/* potential.py:827-836 `regulization` strings */
enum { GH_REG_DAMPING = 0, GH_REG_SMOOTHNESS = 1, GH_REG_MS = 2, GH_REG_TV = 3 };

/* ---- lifetime ------------------------------------------------------------------------- */

/* One inversion problem (N observations, M active cells) on GPU `device`. */
int gh_create(gh_ctx **out, int device, int64_t N, int64_t M);
void gh_destroy(gh_ctx *ctx);
const char *gh_last_error(const gh_ctx *ctx);
/* Device facts for logs/benchmarks: name (<=255 chars), CU count, total memory in bytes. */
int gh_device_info(const gh_ctx *ctx, char *name256, int *cus, int64_t *mem_bytes);

/* ---- sensitivity matrix G (the reference's `kernel` / `A`) ---------------------------- */

/* Observation points: (x, y, z) in m for prisms, (lon, lat, height) in deg/deg/m for
 * tesseroids.  Replaces the xp,yp,zp / lon,lat,height arguments of prism.gz (prism.py:911)
 * and tesseroid.gz (tesseroid.py:421). */
int gh_set_obs(gh_ctx *ctx, const double *a, const double *b, const double *c);
/* Active cells in mesh order, M x 6 row-major: x1,x2,y1,y2,z1,z2 (prism) or
 * w,e,s,n,top,bottom (tesseroid); `ratio` is the tesseroid distance-size ratio
 * (tesseroid.py:77, 1.6 for gz), ignored for prisms. */
int gh_set_cells(gh_ctx *ctx, const double *bounds6, int kind, double ratio);
/* Matrix-free mode (call before gh_build_G): the kernel matrix is never stored; the prism /
 * tesseroid entries are re-evaluated where they are needed.  With N <= 16384 observations a
 * leapfrog step evaluates every entry ONCE (a workgroup keeps a cell's column on the chip between
 * the dot with r, the leapfrog update and the forward accumulation, like the dense sweep); beyond
 * that an adjoint pass and a forward pass evaluate it twice.  For problems whose G exceeds HBM. */
int gh_set_matrix_free(gh_ctx *ctx, int enable);
/* Arithmetic of the matrix-free tesseroid passes (call before gh_build_G; prisms: no effect).
 * exact = 0 (default): a far pair's 2x2x2 Gauss-Legendre leaf re-arranged for throughput --
 * cos(lon - lon') by the addition theorem, 1 / l^3 from the hardware reciprocal square root and one
 * Newton factor; agrees with the reference's values to ~1e-14 (stated tolerance of the path 1e-10).
 * exact = 1: the leaf in the operation order of gravmag/_tesseroid_numba.py:207-222 (same bits as the
 * dense assembly, ~3x the time).  Which pairs are far is decided by the reference's own distance /
 * size test with the same bits either way (_tesseroid_numba.py:135-157).  Without this call the
 * environment variable GRAVHMC_MF_EXACT (0 / 1) chooses, default 0. */
int gh_set_matrix_free_exact(gh_ctx *ctx, int exact);
/* Shift-invariant store for regular spherical grids (call before gh_build_G; implies that G is never
 * stored).  Where every row of tesseroid cells covers the full circle of longitudes with one spacing
 * and the observations sit on that spacing (example/global/main_global.py:25-28: 3-degree mesh under a
 * 3-degree observation grid) the entry of (observation i, cell (row c, longitude k)) depends on the two
 * longitudes through their difference only (gravmag/_tesseroid_numba.py:207-222: cos(lon - lon')):
 * K[i, (c, k)] = T[c][class of (lat_i, h_i)][(m_i - k) mod n].  gh_build_G then evaluates the table T
 * once with the reference's adaptive engine (near-field pairs included; C4: 600 x 61 x 120 doubles =
 * 35 MB instead of 4.25 GB of G or 5.3e8 evaluations per step) and every pass -- forward, adjoint, the
 * fused leapfrog step -- becomes circular correlations along the longitude served from LDS.  Values
 * agree with the stored kernel to the rounding of cos(lon - lon') at a shifted pair of longitudes
 * (stated tolerance 1e-10).  gh_build_G fails with GH_ERR_UNSUPPORTED and the reason when the geometry
 * lacks the structure (or has more than 1024 longitudes per cell row: the transforms' tables live in the LDS).
 * gh_compress_wavelet works on such a context (the compressor's rows are evaluated, the forward runs on the CSR
 * operator, the gradient on the table's adjoint pass); batches of chains are refused with it. */
int gh_set_shift_invariant(gh_ctx *ctx, int enable);
/* The structure gh_build_G found: longitudes per cell row, observation classes (distinct latitude /
 * height pairs), cell rows, bytes of the table (all 0 when the store is not in use). */
int gh_shift_invariant_info(const gh_ctx *ctx, int *n_lon, int *n_classes, int *n_rows, int64_t *table_bytes);
/* The store in the longitude-harmonic domain (csrc/lonsymh.hip.h; the default where the geometry allows it: at
 * most 126 longitudes per cell row and 64 observation classes; GRAVHMC_LONSYM_HARMONIC=0 keeps the direct
 * correlations): a circular correlation along the longitude is a product per frequency of length-n DFTs, so
 * forward and adjoint cost n_rows x n_classes x (n / 2 + 1) complex multiply-adds each instead of N M real ones
 * (C4: 2.2e6 against 5.3e8) from a complex table T^[row][class][frequency] of the same size, read once per
 * leapfrog step; the clamp-and-reflect update (hmc.py:135-141) happens at the longitudes, between an inverse
 * and a forward transform inside the cell row's workgroup.  Grids beyond that form's limits (up to 1024 longitudes,
 * any number of classes -- a 1-degree global grid: 360 longitudes, 181 classes, 1800 cell rows, T^ = 0.96 GB where
 * the dense kernel would be 339 GB) run the same arithmetic as four streaming launches per step that read T^ twice
 * (csrc/lonsymw.hip.h; GRAVHMC_LONSYM_WIDE=0 off, =2 everywhere); on a grid symmetric about the equator that form
 * keeps one row of T^ per north-south mirrored pair of cell rows and lets every entry read serve both (half the bytes;
 * mirrored entries of the reference's kernel agree to rounding, within the stated 1e-10; GRAVHMC_LW_MIRROR=0: off).  on: 0 = direct correlations, 1 = the register
 * form, 2 = the streamed form; frequencies n / 2 + 1, bytes of T^, workgroups
 * of the pass.  gh_batch_* on a shift-invariant context (BASELINE configs[3]: 8 chains) runs every chain as a
 * light context of its own -- stream, chain state, work buffers -- on the shared tables, one host thread per
 * chain; nothing stays in flight between calls (n_started = n_done = T in carry-over mode). */
/* The harmonic pass as ONE persistent launch per gh_chain_run call (csrc/lonres.hip.h): every workgroup keeps its cell
 * rows' part of the table in registers for the whole batch of trajectories and the workgroups exchange the forward
 * partials, the residuals' transforms and the Metropolis sums through memory (bounded waits; a launch that gives up
 * leaves the chain untouched and the call runs on the launches per phase).  All four regularisers; one chain at a time
 * (example/global/run_main.sh:16 runs one chain per rank; the chains of a gh_batch_* batch take turns).  workgroups: grid of the launch (0: not
 * in use); launches, evaluations (forward + adjoint product each), trajectories so far, timeouts.
 * GRAVHMC_LONSYM_RESIDENT=0 switches it off. */
int gh_shift_invariant_resident_stats(gh_ctx *ctx, int *workgroups, int64_t *launches, int64_t *evaluations,
                                      int64_t *trajectories, int *timeouts);
int gh_shift_invariant_harmonic(const gh_ctx *ctx, int *on, int *n_freq, int64_t *table_bytes, int *workgroups);
/* Work of the matrix-free passes since gh_profile_enable(ctx, 1) (fused form only): entries
 * evaluated, 2x2x2 Gauss-Legendre leaves evaluated (tesseroids; = entries for prisms), launches.
 * Tesseroids: the pairs that need the reference's adaptive subdivision (_tesseroid_numba.py:135-157)
 * are a property of the geometry; when they are few (<= 1/64 of the pairs) gh_build_G evaluates
 * them once and keeps them as a sparse near-field list -- near_entries pairs, near_leaves GLQ leaves
 * (0 when the list is not in use and the passes subdivide inside).  Any pointer may be NULL. */
int gh_matrix_free_stats(gh_ctx *ctx, int64_t *entries, int64_t *leaves, int64_t *launches,
                         int64_t *near_entries, int64_t *near_leaves);
/* Assemble the dense kernel on the device.  Replaces the Python cell loop + native calls of
 * prism.py:291-316 -> _prism.pyx:265-290, or tesseroid.py:189-232 -> _tesseroid_numba.py:32-71,
 * including the unit scaling G*SI2MGAL.  For tesseroids returns GH_ERR_OVERFLOW if any
 * (obs, cell) pair needed more than 100 stack entries. */
int gh_build_G(gh_ctx *ctx);
/* Tesseroid diagnostics of the last gh_build_G: number of cells for which the reference
 * would emit its "stopped dividing" RuntimeWarning (tesseroid.py:228-229) and the total
 * number of GLQ leaf evaluations. */
int gh_kernel_stats(const gh_ctx *ctx, int64_t *warn_cells, int64_t *leaves);
/* Explicit dense kernel from the host instead of assembling it (A[i + j*ld] if
 * fortran_order else A[i*ld + j]). */
int gh_upload_G(gh_ctx *ctx, const double *A, int64_t ld, int fortran_order);
/* Copy the device matrix back, column-major with leading dimension ld >= N (what
 * GravMagModule.kernelw() hands out as `Aw`, potential.py:584-589). */
int gh_download_G(gh_ctx *ctx, double *A, int64_t ld);
/* (The `result` output of prism.gz / tesseroid.gz, d = G * rho for the unweighted kernel, is
 * gh_forward called before gh_weight.) */

/* Sensitivity weighting, potential.py:232-264: wm_j = (sum_i G_ij^2)^weightfactor, G <- G *
 * diag(1/wm) in place (columns with zero norm are left untouched, wm_j = 0).  wm_out (M)
 * receives the diagonal of the reference's Wm. */
int gh_weight(gh_ctx *ctx, double weightfactor, double *wm_out);

/* ---- potential (inversion/potential.py:688-845) ---------------------------------------- */

/* Observed data and, optionally, the field of fixed cells (`fixed=True, grav_fix=...`,
 * potential.py:700-703); pass NULL when unused. */
int gh_set_data(gh_ctx *ctx, const double *dobs, const double *grav_fix_or_null);
/* Regulariser: kind (GH_REG_*), alpha (hmc RegulFactor), beta (MS/TV), mesh shape
 * (nz,ny,nx) for Smoothness/TV (must satisfy nz*ny*nx == M), weighted prior
 * mwapr (M).  Needs gh_weight first for MS (uses wm^2 = diag(WmSquare)).
 * On a context that holds a shard of the cells (gh_shard_init*): mwapr is the local part; for
 * Smoothness/TV shape3 is the GLOBAL mesh shape (nz*ny*nx == M_global), every shard must consist
 * of whole z-planes (m0 and M multiples of ny*nx, else GH_ERR_UNSUPPORTED) and the call is
 * collective (the boundary planes of the prior model are exchanged once; those of the model
 * travel with every evaluation's forward partial). */
int gh_set_reg(gh_ctx *ctx, int kind, double alpha, double beta, const int shape3[3],
               const double *mwapr);
/* d = Aw * mw            (potential.py:698, np.dot(self.Aw, mw)) */
int gh_forward(gh_ctx *ctx, const double *mw, double *dpre);
/* g = Aw^T * r           (potential.py:708, np.dot(self.Aw.T, r)) */
int gh_adjoint(gh_ctx *ctx, const double *r, double *g);
/* Drop-in for GravMagModule.misfit_and_grad with constraint='mandatory'
 * (potential.py:812-845): out3 = (misfit, data_value, model_value), grad (M), dpre (N,
 * without grav_fix). */
int gh_misfit_and_grad(gh_ctx *ctx, const double *x, double out3[3], double *grad,
                       double *dpre);

/* Regulariser value and gradient alone (alpha = 1), for optimisers that combine the terms
 * themselves: ConjugateGradient of inversion/reginv.py:271-355.  ms_grad_den_mw = 1 reproduces
 * that file's MS gradient, whose denominator is (mw^2 + beta)^2 (reginv.py:288-292) instead of
 * ((mw - mwapr)^2 + beta)^2 (potential.py:732-735). */
int gh_reg_eval(gh_ctx *ctx, int kind, double beta, const int shape3[3], int ms_grad_den_mw,
                const double *mw, const double *mwapr, double *value, double *grad /* M or NULL */);

/* ---- wavelet-compressed forward operator (gravmag/compressor1D.py, compressor3D.py) ------ */

/* Build the compressed kernel on the device: row-wise db4 / 'periodization' DWT of the weighted
 * kernel (dims = 3: separable over shape3 = (nz,ny,nx) with nz*ny*nx == M; dims = 1: over the
 * flat model, the variant carved meshes need), hard threshold |c| < thr -> 0, CSR with the
 * column layout of pywt.coeffs_to_array.  Replaces compressor3D.kernelcompressor (:17-44) /
 * compressor1D.kernelcompressor (:17-42); the reference uses thr = 1e-3, levels = 2.  After
 * this call gh_misfit_and_grad and the chain evaluate the FORWARD product as
 * Awcp @ DWT(mw) (modelcompressor, compressor3D.py:47-68) while the gradient keeps the exact
 * dense Aw^T (potential.py:693-708).  nnz_out / ncols_out receive the CSR size (N x ncols).
 * Also on a matrix-free or shift-invariant context (the rows are evaluated instead of gathered) and on ROW blocks of
 * a sharded kernel (after gh_shard_init_rows and gh_weight: the compressor works row by row, every rank compresses
 * its own rows -- N, nnz and gh_download_csr are the rank's); refused on column blocks (a row's transform would span
 * the ranks). */
int gh_compress_wavelet(gh_ctx *ctx, int dims, const int shape3[3], double thr, int levels,
                        int64_t *nnz_out, int64_t *ncols_out);
/* The CSR arrays (what GravMagModule.Awcp holds): indptr N+1, indices/data nnz. */
int gh_download_csr(gh_ctx *ctx, int64_t *indptr, int32_t *indices, double *data);
/* Packed wavelet coefficients of a model vector: pywt.coeffs_to_array(wavedec[n](mw))[0]. */
int gh_model_coeffs(gh_ctx *ctx, const double *mw, double *coeff /* ncols */);
/* d = Awcp @ DWT(mw)    (modelcompressor) */
int gh_forward_wavelet(gh_ctx *ctx, const double *mw, double *dpre);

/* ---- HMC chain (inversion/hmc.py:85-177) ----------------------------------------------- */

/* Start (or restart) a chain at weighted model x0 with per-cell bounds low/high (weighted,
 * hmc.py:391-393).  Evaluates the potential at x0 once and keeps x, residual and gradient
 * resident on the device. */
int gh_chain_init(gh_ctx *ctx, const double *x0, const double *low, const double *high);
/* One trajectory of L leapfrog steps with clamp-and-reflect bounds and the Metropolis test
 * (HamitonianMC._leapfrog, hmc.py:85-177).  p0 = randn(M)*Sigma and u = rand() are drawn by
 * the caller in the reference's RNG order (hmc.py:297,95,164).  out5 = (U, U_data, U_model,
 * Hcur, Hnew) with U.. of the state the chain is left in (proposal if accepted, else the
 * starting point).  The device executes one fused sweep of G per leapfrog step (adjoint of
 * step s and forward of step s+1 share the sweep) plus one adjoint-only sweep, which
 * gh_chain_prefetch_momentum can merge into the next trajectory. */
int gh_chain_trajectory(gh_ctx *ctx, const double *p0, double dt, int L, double u,
                        int *accepted, double out5[5]);
/* Announce the momentum of the trajectory AFTER the next gh_chain_trajectory call (same RNG
 * stream, drawn one trajectory ahead).  The last sweep of that call then also takes the
 * announced trajectory's first leapfrog step from the proposal, so an accepted proposal costs L
 * sweeps of G instead of L+1.  Purely an execution-order optimisation: results are bit-identical
 * with and without it; a rejected proposal simply discards the speculative step.  The
 * announcement is consumed by one gh_chain_trajectory call. */
int gh_chain_prefetch_momentum(gh_ctx *ctx, const double *p0_next);
/* K trajectories in one call (the body of HamitonianMC.sample's loop, hmc.py:295-343, without
 * a Python round trip per trajectory): L[k], p0s[k*M..], us[k] in RNG-stream order;
 * p0_lookahead (or NULL) is the momentum of the trajectory after the batch so the speculative
 * first step also crosses batch boundaries.  Accepted proposals are counted since
 * gh_chain_init; the run stops early once `stop_at_accepts` (> 0) have been accepted, and the
 * posterior window (if any) receives every accepted state beyond the first `record_from`
 * (the sampler's ndraws burn-in).  accepted[k], out5s[5k..] as gh_chain_trajectory; x_out (K*M or
 * NULL) receives the chain state after each ACCEPTED trajectory k; n_run = trajectories done. */
int gh_chain_run(gh_ctx *ctx, int K, const int *L, const double *p0s, const double *us, double dt,
                 const double *p0_lookahead, int64_t stop_at_accepts, int64_t record_from,
                 int *accepted, double *out5s, double *x_out, int *n_run);
/* How often the speculative first step was used / discarded. */
int gh_chain_stats(gh_ctx *ctx, int64_t *spec_hits, int64_t *spec_misses);
/* Small dense problems (N <= 1024, one column block per CU fitting its LDS next to the kernel's
 * scratch, one device, stored G): gh_chain_run runs its K trajectories inside ONE
 * launch with G resident in LDS (csrc/resident.hip.h) instead of one sweep per launch.  Same
 * contract and results to rounding (the summation order over the cells differs); environment
 * GRAVHMC_RESIDENT=0 switches it off.  launches / evaluations: how much ran there so far. */
int gh_chain_resident_stats(gh_ctx *ctx, int64_t *launches, int64_t *evaluations);
/* More observations than one workgroup holds of a column (N > 16384, stored G): the fused leapfrog
 * step runs on teams of `members` workgroups that share each column (csrc/teamsweep.hip.h), still
 * ONE read of G per step; adjoint-only / forward-only sweeps and the steps after a team timed out
 * run in row panels.  members: workgroups per team (0: not in use), launches: team sweeps so far,
 * timeouts: launches that gave up (after three the context stays on row panels), late_parts:
 * columns for which some member found its team's parts not yet published when it needed them
 * and had to wait (as of the last synchronisation; columns x members x launches is the total). */
int gh_team_sweep_stats(gh_ctx *ctx, int *members, int64_t *launches, int *timeouts, int64_t *late_parts);
int gh_chain_get_x(gh_ctx *ctx, double *x /* M */);
int gh_chain_get_dsyn(gh_ctx *ctx, double *dsyn /* N, dpre of the current state */);
/* ---- several chains sharing every sweep of G (fp64 MFMA) ---------------------------------- */

/* Up to 16 independent chains (the reference runs them as separate MPI ranks, hmc.py:367-369) on
 * ONE GPU against ONE copy of G: per leapfrog step the adjoint and the forward products of all
 * chains are two skinny GEMMs on v_mfma_f64_16x16x4 (32 flop per byte of G instead of 0.5), so 16
 * chains cost two sweeps of G per step instead of 16.  Each chain keeps its own trajectory
 * length L[c] (chains that finish early idle until the longest is done) and its own Metropolis
 * variate; every chain reproduces what a single-chain context computes from the same inputs.
 * Host arrays are chain-major: x0s, p0s = C rows of M doubles.  Unsharded kernel only; with the
 * wavelet-compressed forward only where the resident chain kernel takes the batch (GH_ERR_UNSUPPORTED
 * otherwise: the MFMA form has no compressed forward).
 * Problems small enough for the resident chain kernel (see gh_chain_resident_stats) do not use
 * the MFMA form, which would be bound by its launches there: the chains take turns inside one
 * launch of that kernel per gh_batch_trajectory call (same contract, same results to rounding). */
int gh_batch_init(gh_ctx *ctx, int C, const double *x0s, const double *low, const double *high);
int gh_batch_trajectory(gh_ctx *ctx, const double *p0s, double dt, const int *L, const double *us,
                        int *accepted /* C */, double *out5s /* C x 5, as gh_chain_trajectory */);
/* Up to T further trajectories of every chain in one call; arrays are chain-major: L[c*T + t],
 * us[c*T + t] and the momentum p0_rows[c*T + t] (pointer to M doubles: the rows need not be
 * contiguous) are the trajectories chain c has not started yet, in order.
 * The chains do not wait for each other: a chain that has finished a trajectory (its L steps and
 * the final half momentum step) is decided and starts its next one in the very next sweep while
 * the others are in the middle of theirs, so every sweep of G carries a step of every chain that
 * has work.  (Rounds of gh_batch_trajectory leave a chain idle from its own L to the longest L of
 * the round: with the reference's Lrange [5,20] and 16 chains 38 % of the sweeps' capacity.)  Each
 * chain computes exactly what it computes through gh_batch_trajectory.
 * Results are reported per chain in order of completion: accepted[c*S + i], out5s[(c*S + i)*5 ..]
 * and, if x_out != NULL, the state after an ACCEPTED trajectory at x_out[(c*S + i)*M ..], for the
 * i-th trajectory chain c COMPLETED in this call; S = T result slots per chain, S = T + 1 in
 * carry-over mode (the trajectory that came in flight plus up to T new ones).
 * n_started == n_done == NULL: all T trajectories of every chain are run to completion.
 * Otherwise (carry-over mode) the call ends as soon as some chain has nothing left to start; the
 * other chains keep their trajectory in flight and continue it in the next call (same dt), where
 * its result is the first that chain reports.  n_started[c] = how many of this call's T chain c
 * has started (pass the rest again, followed by new ones), n_done[c] = completions reported.
 * T = 0 in that mode drains: the trajectories in flight are completed (at most one result per
 * chain).  gh_batch_init discards anything in flight. */
int gh_batch_run(gh_ctx *ctx, int T, const int *L, const double *const *p0_rows, const double *us,
                 double dt, int *accepted, double *out5s, double *x_out, int *n_started, int *n_done);
int gh_batch_get_x(gh_ctx *ctx, int chain, double *x /* M */);
/* Page-locked host memory for momentum rows (the reference draws a trajectory's momentum with
 * np.random.randn(M) * Sigma into a fresh array, inversion/hmc.py:91): rows of gh_batch_run that lie in such a
 * block go to the device straight from it, adjacent rows of a chain's list in one copy; rows anywhere else
 * are gathered into the library's own staging buffer first (gh_batch_staging_stats counts both kinds, lock-step
 * form).  Blocks live until gh_pinned_free or gh_destroy. */
int gh_pinned_alloc(gh_ctx *ctx, size_t bytes, void **host);
int gh_pinned_free(gh_ctx *ctx, void *host);
int gh_batch_staging_stats(gh_ctx *ctx, int64_t *rows_direct, int64_t *rows_staged);
/* Two or more chains on a problem small enough for the resident chain kernel with every column in LDS
 * (<= 32 cells per CU, N <= 640: BASELINE configs[0] and [2]) do not take turns: ALL chains advance in
 * lock-step inside one launch per gh_batch_run / gh_batch_trajectory call (csrc/resbatch.hip.h) -- one
 * three-hop exchange per step of the whole batch instead of one per chain, both products of all chains
 * as fp64 MFMA GEMMs (the reference runs its chains as separate MPI ranks: inversion/hmc.py:367-369; per
 * chain the arithmetic is inversion/hmc.py:85-177 with inversion/potential.py:698,708).  A chain whose
 * trajectory ended takes the first step of its next one speculatively while its Metropolis sums travel; a
 * rejected proposal costs that chain one lock-step.  launches, lock-steps and evaluations of all chains
 * ("chain-steps") so far, lock-steps lost to rejected speculation, timeouts (a launch that gave up waiting
 * for its workgroups: the chains take turns in the resident chain kernel from then on, trajectories in
 * flight are replayed there).  GRAVHMC_RESIDENT_BATCH=0 switches the form off. */
int gh_batch_resident_stats(gh_ctx *ctx, int64_t *launches, int64_t *lock_steps, int64_t *chain_steps,
                            int64_t *lost_steps, int *timeouts);
/* Batched chains on a MATRIX-FREE context (gh_set_matrix_free): every entry a pass evaluates serves all
 * chains (example/global/run_main.sh:16 runs its chains as separate ranks, each re-evaluating the whole
 * tesseroid kernel, gravmag/_tesseroid_numba.py:32-71).  Two forms: an adjoint/update pass and a forward
 * pass (two evaluations per entry and step, any N), or -- while every workgroup of the launch is
 * resident, one per CU -- ONE pass in which the workgroups holding the row chunks of the same column tiles
 * exchange their partial dots through memory (teams; csrc/mfbatch.hip.h).  members x ranges = grid of the
 * team form (0: not in use), launches so far, timeouts: launches that gave up waiting (after the first the
 * context stays on the two-pass form; gh_batch_trajectory repeats the round, gh_batch_run returns an error
 * and wants gh_batch_init again). */
int gh_batch_fused_stats(gh_ctx *ctx, int *members, int *ranges, int64_t *launches, int *timeouts);
/* The same team form for ONE chain on a matrix-free tesseroid context (the leapfrog steps of
 * gh_chain_trajectory / gh_chain_run; N <= 17920 observations, the near-field list in use, the GPU not shared):
 * a wave keeps its column's dot with r as one number per tile, so a team exchanges 16 doubles per member and
 * tile, and the column-per-workgroup phases of the plain pass (constants, slots, barrier, scalars, update,
 * forward) disappear.  A pass that times out (2 s) switches the form off for good; the trajectory is
 * repeated on the column-per-workgroup pass.  Numbers as above. */
int gh_matrix_free_team_stats(gh_ctx *ctx, int *members, int *ranges, int64_t *launches, int *timeouts);

/* Posterior statistics without text I/O.  The reference appends every accepted
 * model as a '%.8f' text row to model.dat (hmc.py:328-332) and its plot scripts take np.mean /
 * np.std over the last 100 rows (plot_uniform.py:44-55,103-104).  gh_posterior_window reserves a
 * ring of the last K models m = WmInv @ mw in HBM; gh_posterior_add stores the chain's current
 * state (call it for every accepted, post-burn-in sample); gh_posterior_read returns the
 * per-cell mean and population standard deviation over the window. */
int gh_posterior_window(gh_ctx *ctx, int K);
int gh_posterior_add(gh_ctx *ctx);
int gh_posterior_read(gh_ctx *ctx, int64_t *n_in_window, int64_t *n_total, double *mean /* M or NULL */,
                      double *sd /* M or NULL */);
/* Stateless convenience with the signature SURVEY 8b lists: init + trajectory + readback. */
int gh_leapfrog(gh_ctx *ctx, double *x_inout, const double *p0, double dt, int L,
                const double *low, const double *high, double u, int *accepted,
                double out5[5], double *dsyn_or_null);

/* ---- one chain sharded over several GPUs -------------------------- */

/* Column-block sharding: rank g of `world` creates its context with its own cell count
 * (gh_create(N, M_local)) and passes only its cells / slices of every M-vector; N-vectors
 * (dobs, grav_fix, dpre) are replicated.  Per potential evaluation the ranks exchange ONE
 * all-reduce of N+2 doubles (forward partial + regulariser partial); the gradient, the
 * regulariser gradient and the leapfrog updates stay local, so the fused one-read sweep is
 * kept.  Damping and MS with any partition; Smoothness / TV with shards of whole z-planes (the
 * boundary planes of the model travel in the same all-reduce, see gh_set_reg).  Every rank ends up
 * with bit-identical scalars, hence identical Metropolis decisions.  (Row blocks -- every rank a
 * slice of the observations -- need the M-vector gradient all-reduced between the adjoint and the
 * update: two reads of the shard per step; gh_shard_init_rows below)
 *
 * RCCL flavour: rank 0 obtains a 128-byte id with gh_shard_unique_id, the launcher
 * broadcasts it, every rank calls gh_shard_init (ncclCommInitRank, collective).  The
 * all-reduce runs on the context's stream over xGMI. */
int gh_shard_unique_id(void *id128);
int gh_shard_init(gh_ctx *ctx, const void *id128, int rank, int world, int64_t M_global, int64_t m0);
/* Host-staged flavour: the all-reduce is delegated to a caller-supplied function that sums
 * `count` doubles in place across ranks (e.g. torch.distributed over gloo); used where RCCL
 * cannot run (several ranks on one GPU) and by the tests. */
typedef int (*gh_allreduce_fn)(void *user, double *host_buf, int64_t count);
int gh_shard_init_callback(gh_ctx *ctx, gh_allreduce_fn fn, void *user, int rank, int world,
                           int64_t M_global, int64_t m0);
/* Row-block sharding, BASELINE configs[4] as it is worded ("G row-block sharded ... RCCL reduce over xGMI for
 * the misfit sum"; SURVEY 8e.2, first form): rank g creates its context with ITS observations
 * (gh_create(N_local, M)), passes only its rows of every N-vector (obs, dobs, grav_fix; dpre comes back for its
 * rows) and FULL model vectors, which are replicated.  Call right after gh_create / gh_build_G, before
 * gh_weight and gh_set_data (both are collective then: a column's norm and the mean of the data span the
 * ranks).  Per potential evaluation: the local rows' forward product, two scalar all-reduces (sum of the
 * predicted data for the mean removal of potential.py:706, |r|^2 -- "the misfit sum") and, because a column's
 * dot with r spans the ranks, an all-reduce of the M-vector gradient between the adjoint pass and the update
 * -- the fused one-read sweep is not possible: TWO reads of the local shard per leapfrog step, and M doubles
 * per step across the ranks instead of the column form's N + 2.  Updates and the regulariser are replicated
 * (any regulariser, no halo).  Same scalars on every rank, hence identical Metropolis decisions.  Stored
 * kernel only; no batches.  gh_compress_wavelet AFTER this call and gh_weight gives every rank the compressed form of its
 * own rows (the forward of the local rows then runs on it). */
int gh_shard_init_rows(gh_ctx *ctx, const void *id128, int rank, int world, int64_t N_global, int64_t n0);
int gh_shard_init_rows_callback(gh_ctx *ctx, gh_allreduce_fn fn, void *user, int rank, int world,
                                int64_t N_global, int64_t n0);
/* Sum of `count` (<= N) doubles in place across the ranks of the shard group (host buffers):
 * lets the host combine per-rank scalars through the same communicator. */
int gh_shard_allreduce(gh_ctx *ctx, double *host_buf, int64_t count);

/* ---- sample files ---------------------------------------------------------------------- */

/* One text row of the reference's model.dat / misfit.dat (np.savetxt(fmt='%.8f', delimiter=' '),
 * hmc.py:241-249): n values as printf("%.8f") would print them, separated by one blank, closed by
 * '\n'.  Host only, no context: at 5*10^5 cells a row is 5.5 MB of text per accepted sample
 *  and np.savetxt's ~85 ns per value would cost as much as the trajectory on the GPU.
 * Every value is rounded from its exact binary expansion like printf does (values whose scaled
 * fraction lies within 1e-7 of a rounding boundary, and non-finite or huge ones, go through
 * snprintf).  Returns the number of bytes written, or -1 if `cap` (>= 24 n + 2 is always enough
 * for |v| < 1e13) is too small. */
int64_t gh_format_row_fixed8(const double *v, int64_t n, char *out, int64_t cap);

/* ---- the reference's random stream ------------------------------------------------------ */

/* NumPy's legacy RandomState stream on the host, bit for bit (MT19937; doubles from two outputs;
 * polar Gaussian with its cached second value; masked-rejection integers): what the reference
 * draws per trajectory in HamitonianMC.sample / _leapfrog -- L = np.random.randint(Lmin, Lmax + 1)
 * (hmc.py:297), p0 = np.random.randn(M) * Sigma (hmc.py:95), u = np.random.rand() (hmc.py:164) --
 * K trajectories per call, written straight into the arrays gh_chain_run takes.  Host only, no
 * context.  gh_rng_create(seed) is np.random.seed(seed); get / set_state exchange the stream with
 * np.random.get_state() / set_state() (key[624], pos, has_gauss, cached_gaussian). */
typedef struct gh_rng gh_rng;
int gh_rng_create(gh_rng **out, uint32_t seed);
void gh_rng_destroy(gh_rng *rng);
int gh_rng_set_state(gh_rng *rng, const uint32_t *key624, int pos, int has_gauss, double cached);
int gh_rng_get_state(const gh_rng *rng, uint32_t *key624, int *pos, int *has_gauss, double *cached);
/* threads of a draw: 1 = the calling thread alone, 2 = the scale pass (the logarithms) on a second thread behind the
 * sequential generation; 0 = default (GRAVHMC_RNG_THREADS, else 2); -1 = 2 where the process has four cores or more */
int gh_rng_set_threads(gh_rng *rng, int threads);
/* host cores this process may use: the affinity mask capped by the cgroup's CPU quota */
int gh_host_cores(void);
int gh_rng_draw_trajectories(gh_rng *rng, int K, int Lmin, int Lmax, int64_t M, double sigma, int *L,
                             double *p0s /* K x M */, double *us /* K */);

/* ---- measurement ----------------------------------------------------------------------- */

/* HIP-event timing of the G sweeps (the dominant kernel) on the context's stream.
 * enable != 0 starts recording (and clears the counters).  gh_profile_read sums the recorded
 * intervals of the launches that read the most bytes of G (where one-read sweeps and row-panel
 * launches mix, the one-read sweeps alone: one kernel, comparable with a kernel trace): total
 * milliseconds, number of those launches, bytes of G each of them reads. */
int gh_profile_enable(gh_ctx *ctx, int enable);
int gh_profile_read(gh_ctx *ctx, double *sweep_ms, int64_t *sweep_launches,
                    int64_t *bytes_per_sweep);
/* Attainable streaming-read rate of this device on this context's stored G (the figure SURVEY
 * 8d asks to report next to the nominal 8 TB/s): `reps` timed passes of a read-only kernel over
 * the whole matrix with the sweep's load form (16-byte loads, non-temporal for nt != 0, four
 * independent loads in flight per thread), best of three launch shapes; milliseconds per pass. */
int gh_measure_stream_read(gh_ctx *ctx, int nt, int reps, double *ms_per_pass);
/* Block until everything queued on the context's stream has finished. */
int gh_synchronize(gh_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* GRAVHMC_H */
