// Resident batch kernel (gfx950): C chains (C <= 16) in LOCK-STEP on a sensitivity matrix that lives on the
// chip -- BASELINE configs[0] / north_star's target configuration (600 x 6000) and, through the dense form
// of the compressed forward, configs[2] -- one launch per call of gh_batch_run.
//
// The single-chain resident kernel (resident.hip.h) spends ~3 of its 6.5 us per evaluation in three
// dependent hops between workgroups; chains that take turns in it pay those hops once per chain.  Here one
// LOCK-STEP advances every chain by one potential evaluation (potential.py:688-845) and one leapfrog update
// (hmc.py:114-152), and all chains travel through the same three hops:
//
//   forward      D (N x 16) = G_w (N x cpw) . X (cpw x 16): the workgroup's columns from LDS, the chains'
//                positions as the B operand of v_mfma_f64_16x16x4; the partial goes straight from the
//                accumulators to memory as tagged granules, rows x chains
//   hops 1-3     as in resident.hip.h (cluster = workgroups with equal w % 8: one XCD; row chunk owners;
//                cluster sums through L2, one cross-XCD hop, result back through L2), each thread owning a
//                (row, chain) pair; granule pairs move as ONE 16-byte access
//   scalars      the three sums a Metropolis test needs (R, p'p after, p'p before) ride the same exchange as
//                four extra rows of the vector -- no separate gather
//   residual     every wave reads ITS rows of d for all chains directly in MFMA operand order, mean removal
//                (potential.py:700-706) through one LDS reduction
//   adjoint      S (cpw x 16) = G_w^T . R: the wave's rows of the workgroup's columns sit in registers
//                (A operand), its rows of the residuals are the B operand; the eight waves' partial
//                products meet in LDS
//   update       one thread per (cell, chain): gradient, momentum / position update with clamp-and-reflect;
//                the thread keeps its cell's chain state (x, p, current sample and its gradient) in registers
//
// Trajectories.  Chain c follows its own list of trajectories (L, momentum, Metropolis variate), all chains
// desynchronised: a chain whose evaluation was the last of its trajectory takes the final half momentum
// step, and in the NEXT lock-step its three sums travel with the exchange while it already takes the first
// step of its next trajectory from the proposal (speculating on acceptance, as gh_batch_run's MFMA form
// does: hmc.py:158-177 decided by every workgroup identically from the exchanged sums).  Accepted: the
// evaluation that arrived is the new trajectory's first.  Rejected: the chain starts again from its current
// sample (one lock-step of this chain lost).  A chain without a next trajectory gets its decision without
// the speculative step.  With `stop_any` (carry-over mode of gh_batch_run) the launch ends as soon as a
// chain has nothing left to start and no decision is pending; trajectories in flight are saved (position,
// momentum, step count) and continue in the next launch.
//
// Every wait is bounded (resident.hip.h's res_poll); on a time-out every workgroup leaves without writing
// chain state and the host continues on the chains-take-turns kernel.
#pragma once
#include "resident.hip.h"
#include "batch.hip.h"

namespace ghk {

constexpr int RB_THREADS = 512;
constexpr int RB_WAVES = 8;
constexpr int RB_XROWS = 4;   // extra rows of an exchanged vector: R share, p'p after, p'p before, spare
constexpr int RB_CST = 8;     // doubles of per-chain state kept between launches

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct ResBatchArgs {
    const double *G;   // Aw: adjoint operand (registers)
    const double *Gl;  // forward operand (LDS): Aw, or the dense model-space form of the compressed forward
    int64_t ld, N, M;
    int cols_per_wg, nwg, try_local;
    const double *gfix, *dobs_c, *low, *high;
    int kind, nz, ny, nx;
    double alpha, beta;
    const double *mwapr, *wm2;
    int C;             // chains
    int T;             // trajectories offered per chain; element (c, t) of L / us at c * T + t
    const int *L;
    const double *p0s; // (C T) x M momenta, chain-major like the lists: row c * T + t
    const double *us;
    double dt;
    int stop_any;      // 1: end when a chain has nothing left to start (carry-over); 0: run all lists to their end
    // chain state, kept between launches
    double *x_cur;     // C x M current samples
    double *g_cur;     // C x M full gradient there
    double *u_cur;     // C x 3 {U, U_data, R}
    double *xs_io;     // C x M position of the trajectory in flight
    double *ps_io;     // C x M momentum
    double *pst_io;    // C x M momentum the trajectory in flight started with (to replay it elsewhere)
    double *cst_io;    // C x RB_CST {in flight, steps done, L, u, p'p before, -, -, -}
    // results: slot i of chain c at c * Tout + i, in the order the chain's trajectories end
    int Tout;
    int *accepted;
    double *out5s;
    double *xacc;      // (C Tout) x M accepted models, or nullptr
    int *n_io;         // [c] list elements started, [16 + c] results, [32] lock-steps, [48 + c] lock-steps chain c
                       // lost to a rejected speculation, [64 + c] trajectory in flight, [80 + c] evaluations of chain c
    // exchange buffers (tags continue across launches)
    double *slabd;     // nwg x ldx x 16 forward partials + scalar rows of the workgroups (chain k4 + 4 q of a row at
                       // position 4 k4 + q), published by ONE flag word per workgroup and lock-step
    u64 *flagg;        // nwg: tag of the lock-step whose partial is complete in slabd
    u32x4 *xslabg;     // 2 x 8 x ldx x C cluster sums, double-buffered by lock-step parity
    u32x4 *dclg;       // 8 x ldx x C finished sums, one copy per cluster
    u64 *xccg;         // nwg {launch tag, XCC id}
    double *xpub;      // 2 x C x M models as the stencil regularisers see them
    unsigned tag0, ltag;
    unsigned *abort_w;
    long long *dbg;    // optional 2 x 16 phase clocks
};

// LDS column stride.  Adjoint operand in registers (GREG): the two column groups a ds_read_b64 half-wave of
// the forward covers must fall into different halves of the 64 banks -> stride = 16 (mod 32) doubles.
// (one copy of G serving both products: stride = 18 (mod 32) -- the adjoint's operand reads, 16 columns x 2 rows
// per half-wave, are then conflict-free, the forward's, 2 columns x 16 rows, 2-way at most)
__host__ __device__ inline int rb_ldp(int ld, bool greg)
{
    const int want = greg ? 16 : 18;
    return ld + ((want - ld % 32) + 32) % 32;
}

// (the waves' partial adjoint products: one slot of 64 doubles per wave and group of 4 columns; the fixed
// part of the data term only where there is one)
static inline size_t resbatch_lds_doubles(int64_t ld, int cols_per_wg, bool have_fix, bool greg)
{
    const size_t cpw4 = ((size_t)cols_per_wg + 3) & ~(size_t)3;
    return cpw4 * (size_t)rb_ldp((int)ld, greg) + (size_t)RB_WAVES * (cpw4 / 4) * 64 + cpw4 * 16 + (have_fix ? 2 : 1) * (size_t)ld + 128 +
           128 + 384 + 64 + 2 + 16;
}

__device__ __forceinline__ u32x4 rb_pack(unsigned tag, double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    return u32x4{(unsigned)b, tag, (unsigned)(b >> 32), tag};
}

__device__ __forceinline__ double rb_value(u32x4 w)
{
    return __longlong_as_double((long long)(((u64)w.z << 32) | (u64)w.x));
}

// one granule pair, 16 bytes: write-through (visible to every XCD) or plain (stays in this XCD's L2)
__device__ __forceinline__ void rb_store(__amdgpu_buffer_rsrc_t rs, unsigned off, u32x4 w, bool local)
{
    if (local)
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)off, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)off, 0, 16);  // sc1
}

// two doubles, 16 bytes, untagged (the forward partials: complete once their workgroup's flag carries the tag)
__device__ __forceinline__ void rb_store2(__amdgpu_buffer_rsrc_t rs, unsigned off, double x, double y, bool local)
{
    const u64 bx = (u64)__double_as_longlong(x), by = (u64)__double_as_longlong(y);
    const u32x4 w = u32x4{(unsigned)bx, (unsigned)(bx >> 32), (unsigned)by, (unsigned)(by >> 32)};
    if (local)
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)off, 0, 0);
    else
        __builtin_amdgcn_raw_buffer_store_b128(w, rs, (int)off, 0, 16);  // sc1
}

__device__ __forceinline__ d2 rb_load2(__amdgpu_buffer_rsrc_t rs, unsigned off)
{
    const u32x4 w = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 16);  // sc1: L2-served
    return d2{__longlong_as_double((long long)(((u64)w.y << 32) | (u64)w.x)),
              __longlong_as_double((long long)(((u64)w.w << 32) | (u64)w.z))};
}

// NB granule pairs per lane (offsets off[u], valid while u < n), all requested together with sc1 loads
// (L2-served), re-read until every one carries `tag`; values into v.  false: timed out / aborted.
template <int NB>
__device__ __forceinline__ bool rb_poll(unsigned *abort_w, __amdgpu_buffer_rsrc_t rs, unsigned tag, int n,
                                        const unsigned (&off)[NB], double (&v)[NB])
{
    bool have[NB];
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        have[u] = !(u < n);
        v[u] = 0.0;
    }
    unsigned spins = 0;
    long long t0 = 0;
    for (;;) {
        asm volatile("" ::: "memory");
        u32x4 wv[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (!have[u]) wv[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off[u], 0, 16);  // sc1
        bool ok = true;
#pragma unroll
        for (int u = 0; u < NB; ++u)
            if (!have[u]) {
                if (wv[u].y == tag && wv[u].w == tag) {
                    have[u] = true;
                    v[u] = rb_value(wv[u]);
                } else {
                    ok = false;
                }
            }
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u || now - t0 > RES_TIMEOUT_TICKS) {
                __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// KS: k-steps (4 rows each) a wave contracts in the adjoint: 32 KS >= ld.  NT: 16-column tiles of the
// workgroup's columns (cols_per_wg <= 16 NT).  GREG: the adjoint's operand (the wave's rows of the columns of
// a.G) lives in registers -- needed when the forward operator in LDS is another matrix (compressed forward);
// otherwise both products read the ONE copy in LDS (80 VGPRs less: no spills; an MFMA of 64 cycles needs 512
// bytes of it, the LDS delivers 256 per cycle).
template <int KS, int NT, bool GREG>
__global__ void __launch_bounds__(RB_THREADS) resident_batch_kernel(ResBatchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lo4 = lane & 15, k4 = lane >> 4;
    const int w = blockIdx.x;
    const int ld = (int)a.ld, ldx = ld + RB_XROWS;
    const int cpw = a.cols_per_wg, cpw4 = (cpw + 3) & ~3, KF = cpw4 >> 2;
    const int ldp = rb_ldp(ld, GREG);
    const int64_t M = a.M;
    const int64_t j0 = (int64_t)w * cpw;
    const int nc = (int)((M - j0 < cpw) ? (M - j0) : cpw);
    const int nwg = a.nwg, C = a.C, T = a.T, Tout = a.Tout;
    const bool stencil = (a.kind == 1 || a.kind == 3);
    const int N = (int)a.N;

    double *Gs = smem;                               // cpw4 x ldp
    double *part = Gs + (size_t)cpw4 * ldp;          // 8 x KF x 64 partial adjoint products (groups of 4 columns)
    double *Xs = part + RB_WAVES * KF * 64;          // cpw4 x 16 positions, B operand of the forward
    double *dcs = Xs + cpw4 * 16;                    // ld
    double *gfs = dcs + ld;                          // ld, only with a fixed part of the data term
    double *redm = gfs + (a.gfix ? ld : 0);          // 8 x 16
    double *redu = redm + 128;                       // 8 x 16
    double *shw = redu + 128;                        // 8 x 3 x 16 shares of the waves' cells
    double *tots = shw + 384;                        // 3 x 16 exchanged sums (+ 16 spare)
    int *flag_s = reinterpret_cast<int *>(tots + 64);
    long long *tacc_s = reinterpret_cast<long long *>(tots + 64 + 2);

    long long tlast = 0;
    const bool timing = a.dbg != nullptr && tid == 0 && (w == 0 || w == nwg - 1);
    auto tick = [&](int slot) {
        if (timing) {
            const long long now = wall_clock64();
            tacc_s[slot] += now - tlast;
            tlast = now;
        }
    };
    if (timing) {
        for (int i = 0; i < 16; ++i) tacc_s[i] = 0;
        tlast = wall_clock64();
    }

    // ---- resident operands
    for (int e = tid; e < cpw4 * ld; e += RB_THREADS) {
        const int c = e / ld, i = e - c * ld;
        Gs[(size_t)c * ldp + i] = (c < nc) ? __builtin_nontemporal_load(a.Gl + (j0 + c) * a.ld + i) : 0.0;
    }
    for (int e = tid; e < cpw4 * 16; e += RB_THREADS) Xs[e] = 0.0;
    for (int i = tid; i < ld; i += RB_THREADS) {
        dcs[i] = (i < N) ? a.dobs_c[i] : 0.0;
        if (a.gfix) gfs[i] = (i < N) ? a.gfix[i] : 0.0;
    }
    if (tid == 0) *flag_s = 1;
    const int rb0 = wave * 4 * KS;  // first row of the wave's share of the adjoint
    double gt[GREG ? NT : 1][GREG ? KS : 1];
    if (GREG) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int col = 16 * t + lo4, row = rb0 + 4 * s + k4;
                gt[GREG ? t : 0][GREG ? s : 0] = (col < nc && row < ld) ? a.G[(j0 + col) * a.ld + row] : 0.0;
            }
    }

    // ---- clusters and placement (resident.hip.h)
    const int ncl = nwg < RES_CLUSTERS ? nwg : RES_CLUSTERS;
    const int cg = w % RES_CLUSTERS, crank = w / RES_CLUSTERS;
    const int cn = (nwg - cg + RES_CLUSTERS - 1) / RES_CLUSTERS;
    const int nch = (nwg / RES_CLUSTERS) < 1 ? 1 : nwg / RES_CLUSTERS;
    const int ch = (ldx + nch - 1) / nch;
    bool local = false;
    {
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID[3:0]
        if (tid == 0)
            __hip_atomic_store(a.xccg + w, ((u64)a.ltag << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();  // flag_s, LDS images
        if (wave == 0) {
            bool same = true;
            const bool got = res_poll(a.abort_w, [&]() -> bool {
                bool ok = true;
                if (lane < cn) {
                    const u64 e = __hip_atomic_load(a.xccg + cg + RES_CLUSTERS * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (unsigned)(e >> 32) == a.ltag;
                    same = ((unsigned)e & 0xfu) == xcc;
                }
                return ok;
            });
            if (lane == 0) tots[48] = (got && __all(same) && a.try_local) ? 1.0 : 0.0;
            if (!got) *flag_s = 0;
        }
        __syncthreads();
        local = tots[48] != 0.0;
        if (*flag_s == 0) return;
    }

    const __amdgpu_buffer_rsrc_t rs_slab =
        __builtin_amdgcn_make_buffer_rsrc(a.slabd, 0, (int)((size_t)(nwg + 8) * ldx * 16 * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xs = __builtin_amdgcn_make_buffer_rsrc(a.xslabg, 0, 2 * RES_CLUSTERS * ldx * C * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dcl = __builtin_amdgcn_make_buffer_rsrc(a.dclg, 0, RES_CLUSTERS * ldx * C * 16, 0x00020000);

    // ---- chain state of this thread: cell jt of the workgroup, chain ct.  The per-chain scalars are kept by
    // EVERY thread for the chain its lane feeds in the MFMA operand layouts (lane & 15 = tid & 15).
    const int ct = lo4, jt = tid >> 4;
    const bool cell = jt < nc && ct < C;
    const int64_t jg = j0 + jt;
    double xs = 0.0, ps = 0.0, xc = 0.0, gc = 0.0, xprop = 0.0, gprop = 0.0, p0k = 0.0, p0n = 0.0;
    double clo = 0.0, chi = 0.0, apr = 0.0, w2 = 1.0;
    enum { IDLE = 0, RUN = 1, DECIDE = 2 };
    int mode = IDLE, s_done = 0, Lc = 0, q_next = 0, n_done = 0;
    bool spec = false;
    double uc = 0.0, pp0 = 0.0, udL = 0.0, U0 = 0.0, U1 = 0.0, U2 = 0.0;
    int Ln = 0;        // length and variate of the next list element (fetched ahead, with its momentum)
    double un = 0.0;
    double sh_r = 0.0, sh_pp1 = 0.0, sh_pp0 = 0.0;  // this cell's terms of the three sums, for the next exchange
    if (ct < C) {
        U0 = a.u_cur[3 * ct];
        U1 = a.u_cur[3 * ct + 1];
        U2 = a.u_cur[3 * ct + 2];
        if (a.cst_io[RB_CST * ct] != 0.0) {
            mode = RUN;
            s_done = (int)a.cst_io[RB_CST * ct + 1];
            Lc = (int)a.cst_io[RB_CST * ct + 2];
            uc = a.cst_io[RB_CST * ct + 3];
            pp0 = a.cst_io[RB_CST * ct + 4];
        }
    }
    if (cell) {
        xc = a.x_cur[(int64_t)ct * M + jg];
        gc = a.g_cur[(int64_t)ct * M + jg];
        clo = a.low[jg];
        chi = a.high[jg];
        apr = a.mwapr[jg];
        if (a.kind == 2) w2 = a.wm2[jg];
        if (mode == RUN) {
            xs = a.xs_io[(int64_t)ct * M + jg];
            ps = a.ps_io[(int64_t)ct * M + jg];
            p0k = a.pst_io[(int64_t)ct * M + jg];
        }
    }
    RegArgs ra{};
    ra.ms_grad_den_mw = 0;
    ra.kind = a.kind;
    ra.M = M;
    ra.nz = a.nz;
    ra.ny = a.ny;
    ra.nx = a.nx;
    ra.alpha = a.alpha;
    ra.beta = a.beta;
    ra.mwapr = a.mwapr;
    ra.wm2 = a.wm2;

    // first step of list element q of chain ct from (x0, g0): hmc.py:95-113
    auto start_from = [&](double x0, double g0, double p0) {
        double pj = p0 - 0.5 * a.dt * g0;
        double xj = x0 + a.dt * pj;
        if (xj > chi) {
            xj = chi;
            pj = -pj;
        } else if (xj < clo) {
            xj = clo;
            pj = -pj;
        }
        ps = pj;
        xs = xj;
    };
    auto list_p0 = [&](int q) -> double {
        return (cell && q < T) ? a.p0s[((int64_t)ct * T + q) * M + jg] : 0.0;
    };
    // a chain with nothing in flight starts the first element of its list
    if (ct < C && mode == IDLE && T > 0) {
        mode = RUN;
        s_done = 0;
        Lc = a.L[ct * T];
        uc = a.us[ct * T];
        q_next = 1;
        p0k = list_p0(0);
        if (cell) {
            start_from(xc, gc, p0k);
            sh_pp0 = p0k * p0k;
        }
    }
    // The next list element (momentum, length, variate) is fetched two evaluations before the running
    // trajectory ends: early enough to hide the latency of the fetch, and nothing a lock-step waits for.
    // (Sending the later rows on a copy stream BESIDE the running kernel was tried: a copy the runtime performs
    // with a shader cannot start while this kernel holds every CU's registers -- the kernel then waits for rows
    // that wait for the kernel.  All rows are in place before the launch.)
    bool fetched = false;
    auto fetch_next = [&]() {
        p0n = list_p0(q_next);
        Ln = (ct < C && q_next < T) ? a.L[ct * T + q_next] : 0;
        un = (ct < C && q_next < T) ? a.us[ct * T + q_next] : 0.0;
        fetched = true;
    };

    // the cells' terms of the three sums -> per-wave shares in LDS (summed over the waves by the next forward)
    auto post_shares = [&]() {
        double v0 = cell ? sh_r : 0.0, v1 = cell ? sh_pp1 : 0.0, v2 = cell ? sh_pp0 : 0.0;
        v0 += __shfl_xor(v0, 16, 64);
        v0 += __shfl_xor(v0, 32, 64);
        v1 += __shfl_xor(v1, 16, 64);
        v1 += __shfl_xor(v1, 32, 64);
        v2 += __shfl_xor(v2, 16, 64);
        v2 += __shfl_xor(v2, 32, 64);
        if (lane < 16) {
            shw[(wave * 3 + 0) * 16 + lane] = v0;
            shw[(wave * 3 + 1) * 16 + lane] = v1;
            shw[(wave * 3 + 2) * 16 + lane] = v2;
        }
    };
    if (cell) Xs[jt * 16 + ct] = xs;
    post_shares();

    int step = 0, lost = 0, evals = 0;
    for (;; ++step) {
        const unsigned tag = a.tag0 + (unsigned)step + 1u;
        const int par = step & 1;
        // ---- stop?  (uniform: every wave holds all chains in its lanes 0..15)
        {
            const u64 m16 = 0xffffull;
            const u64 b_run = __ballot(mode == RUN) & m16, b_dec = __ballot(mode == DECIDE) & m16;
            const u64 b_starved = __ballot(ct < C && mode == IDLE && q_next >= T) & m16;
            const bool none_active = (b_run | b_dec) == 0;
            if (none_active || (a.stop_any && T > 0 && b_starved != 0 && b_dec == 0)) break;
        }
        if (!fetched && (mode != RUN || Lc - s_done <= 2)) fetch_next();
        // (stencil regularisers: the neighbours read this model once they have d, which depends on a partial
        // of every workgroup -- the stores are drained, behind the forward products that hide their latency,
        // before any partial of this workgroup is published)
        if (stencil && cell) st_wt(a.xpub + ((int64_t)par * C + ct) * M + jg, xs);
        __syncthreads();  // Xs, shw complete
        tick(0);
        // ---- forward of all chains: row tiles wave, wave + 8, ...
        {
            constexpr int RT = (2 * KS + RB_WAVES - 1) / RB_WAVES;  // row tiles per wave
            double bx[8];
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) bx[ks] = (ks < KF) ? Xs[(4 * ks + k4) * 16 + lo4] : 0.0;
            d4 acc[RT];
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int rt = wave + RB_WAVES * i;
                acc[i] = d4{0.0, 0.0, 0.0, 0.0};
                if (16 * rt < ld) {
                    double av[8];
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        av[ks] = (ks < KF) ? Gs[(size_t)(4 * ks + k4) * ldp + 16 * rt + lo4] : 0.0;
                    __builtin_amdgcn_sched_barrier(0);
                    // (D'[chain][row] = sum_k X[k][chain] G[row][k]: the lane of row lo4 gets the chains k4 + 4 q,
                    // four neighbours of the slab's row -> two 16-byte stores)
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks)
                        if (ks < KF) acc[i] = mfma_f64(bx[ks], av[ks], acc[i]);
                }
            }
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const int rt = wave + RB_WAVES * i;
                if (16 * rt < ld) {
                    const unsigned off = (unsigned)((((size_t)w * ldx + 16 * rt + lo4) * 16 + 4 * k4) * 8);
                    rb_store2(rs_slab, off, acc[i][0], acc[i][1], local);
                    rb_store2(rs_slab, off + 16, acc[i][2], acc[i][3], local);
                }
            }
            if (wave == 0) {
                // the scalar rows: the three sums' shares of this workgroup's cells (row 3 is spare: zero)
                const int q = lane >> 4;
                double sv = 0.0;
                if (q < 3) {
#pragma unroll
                    for (int v = 0; v < RB_WAVES; ++v) sv += shw[(v * 3 + q) * 16 + lo4];
                }
                const unsigned off = (unsigned)((((size_t)w * ldx + ld + q) * 16 + (lo4 & 3) * 4 + (lo4 >> 2)) * 8);
                const u64 b = (u64)__double_as_longlong(sv);
                typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
                const u32x2 w2v = u32x2{(unsigned)b, (unsigned)(b >> 32)};
                if (local)
                    __builtin_amdgcn_raw_buffer_store_b64(w2v, rs_slab, (int)off, 0, 0);
                else
                    __builtin_amdgcn_raw_buffer_store_b64(w2v, rs_slab, (int)off, 0, 16);
            }
            // the partial is complete in memory (for the stencil regularisers: the model too) -> ONE flag word
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                // (only the workgroups of the own cluster read it: through the shared L2 where the placement allows)
                if (local)
                    __hip_atomic_store(a.flagg + w, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                else
                    __hip_atomic_store(a.flagg + w, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        tick(1);
        // ---- hops 1 + 2: the owner of row chunk `crank` sums it over its cluster, then over the clusters
        if (crank < nch) {
            // the members' flags: every wave looks for itself, its loads of their partials follow its own poll
            {
                const bool got = res_poll(a.abort_w, [&]() -> bool {
                    bool ok = true;
                    if (lane < cn)
                        ok = __hip_atomic_load(a.flagg + cg + RES_CLUSTERS * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (u64)tag;
                    return ok;
                });
                if (!got) *flag_s = 0;
                asm volatile("" ::: "memory");
            }
            tick(2);
            const int r0 = crank * ch;
            const int nr = (ldx - r0 < ch) ? (ldx - r0) : ch;
            const int nitems = nr > 0 ? nr * 8 : 0;   // (row, pair of slab positions)
            const int mt = (cn + 2) / 3;              // members per third of the cluster
            // work unit = item x third of the members: lanes 3 i, 3 i + 1, 3 i + 2 of a wave (lane 63 idle)
            for (int base = 0; base < nitems; base += 21 * RB_WAVES) {
                const int it = base + wave * 21 + lane / 3, g = lane % 3;
                const bool act = lane < 63 && it < nitems;
                const int row = r0 + (act ? it >> 3 : 0), h = it & 7;
                const unsigned rp = (unsigned)((((size_t)row) * 16 + 2 * h) * 8);
                d2 v[11];
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    const int m = g * mt + u;
                    v[u] = d2{0.0, 0.0};
                    if (act && u < mt && m < cn) v[u] = rb_load2(rs_slab, (unsigned)((size_t)(cg + RES_CLUSTERS * m) * ldx * 16 * 8) + rp);
                }
                d2 sg = d2{0.0, 0.0};
#pragma unroll
                for (int u = 0; u < 11; ++u) {
                    sg.x += v[u].x;
                    sg.y += v[u].y;
                }
                // the thirds in a fixed order, (0 + 1) + 2; lane g = 0 of the item carries on with the chain at slab
                // position 2 h, lane g = 1 with the one at 2 h + 1 (position p holds chain p / 4 + 4 (p % 4))
                const double up_x = __shfl(sg.x, lane + 1, 64), up2_x = __shfl(sg.x, lane + 2, 64);
                const double dn_y = __shfl(sg.y, lane - 1, 64), up_y = __shfl(sg.y, lane + 1, 64);
                const double csum = g == 0 ? (sg.x + up_x) + up2_x : (dn_y + sg.y) + up_y;
                tick(3);
                const int ps = 2 * h + g, chn = ps / 4 + 4 * (ps % 4);
                const bool own = act && g < 2 && chn < C;
                const unsigned xoff = (unsigned)((size_t)par * RES_CLUSTERS * ldx * C * 16);
                const unsigned rc = (unsigned)(((size_t)row * C + (own ? chn : 0)) * 16);
                if (own) rb_store(rs_xs, xoff + (unsigned)((size_t)cg * ldx * C * 16) + rc, rb_pack(tag, csum), false);
                unsigned off[8];
                double vv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) off[u] = xoff + (unsigned)((size_t)u * ldx * C * 16) + rc;
                if (!rb_poll<8>(a.abort_w, rs_xs, tag, own ? ncl : 0, off, vv)) *flag_s = 0;
                double tot = 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) tot += vv[u];
                if (own) rb_store(rs_dcl, (unsigned)((size_t)cg * ldx * C * 16) + rc, rb_pack(tag, tot), local);
                tick(4);
            }
        }
        // ---- hop 3: this wave's rows of d for all chains, in B-operand order; wave 0 also the three sums
        double rf[KS];
        {
            constexpr int H = (KS + 1) / 2;
            const unsigned base = (unsigned)((size_t)cg * ldx * C * 16);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned off[H];
                double v[H];
                int n = 0;
#pragma unroll
                for (int u = 0; u < H; ++u) {
                    const int s = h * H + u, row = rb0 + 4 * s + k4;
                    off[u] = base + (unsigned)(((size_t)(row < ld ? row : 0) * C + lo4) * 16);
                    if (s < KS && rb0 + 4 * s < ld) n = u + 1;  // (uniform over the wave; rows past ld inside the last k-step read row 0 and are zeroed below)
                }
                if (lo4 >= C) n = 0;
                if (!rb_poll<H>(a.abort_w, rs_dcl, tag, n, off, v)) *flag_s = 0;
#pragma unroll
                for (int u = 0; u < H; ++u) {
                    const int s = h * H + u;
                    if (s < KS) rf[s] = (rb0 + 4 * s + k4 < ld) ? v[u] : 0.0;
                }
            }
            if (wave == 0) {
                unsigned off[1];
                double v[1];
                const int q = lane >> 4;
                off[0] = base + (unsigned)(((size_t)(ld + (q < 3 ? q : 0)) * C + lo4) * 16);
                if (!rb_poll<1>(a.abort_w, rs_dcl, tag, (q < 3 && lo4 < C) ? 1 : 0, off, v)) *flag_s = 0;
                if (q < 3) tots[q * 16 + lo4] = v[0];
            }
        }
        tick(6);
        // ---- mean removal, residual, data misfit of every chain (potential.py:700-706)
        {
            double sm = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int row = rb0 + 4 * s + k4;
                if (row < ld) {
                    if (a.gfix) rf[s] += gfs[row];
                    if (row < N) sm += rf[s];
                }
            }
            sm += __shfl_xor(sm, 16, 64);
            sm += __shfl_xor(sm, 32, 64);
            if (lane < 16) redm[wave * 16 + lane] = sm;
        }
        __syncthreads();
        if (*flag_s == 0) return;  // (uniform: every wait of this lock-step lies before this barrier)
        double ud = 0.0;
        {
            double mean = 0.0;
#pragma unroll
            for (int v = 0; v < RB_WAVES; ++v) mean += redm[v * 16 + lo4];
            mean /= (double)N;
            double acc = 0.0;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int row = rb0 + 4 * s + k4;
                double r = 0.0;
                if (row < N) {
                    r = (rf[s] - mean) - dcs[row];
                    acc += r * r;
                }
                rf[s] = r;
            }
            acc += __shfl_xor(acc, 16, 64);
            acc += __shfl_xor(acc, 32, 64);
            if (lane < 16) redu[wave * 16 + lane] = acc;
        }
        tick(7);
        // ---- adjoint of all chains: the wave's rows, partial products to LDS
        {
            d4 acc[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = d4{0.0, 0.0, 0.0, 0.0};
            if (GREG) {
#pragma unroll
                for (int s = 0; s < KS; ++s)
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = mfma_f64(gt[GREG ? t : 0][GREG ? s : 0], rf[s], acc[t]);
            } else {
                // operand from the LDS copy: column 16 t + lo4 (zero columns beyond cpw4 are not stored: clamp and
                // mask), rows of the k-step; five k-steps of reads in flight in front of their MFMAs
                const int c0 = lo4 < cpw4 ? lo4 : 0, c1 = 16 + lo4 < cpw4 ? 16 + lo4 : 0;
                const bool z0 = lo4 >= cpw4, z1 = 16 + lo4 >= cpw4;
                const double *g0 = Gs + (size_t)c0 * ldp + rb0 + k4, *g1 = Gs + (size_t)c1 * ldp + rb0 + k4;
#pragma unroll
                for (int sb = 0; sb < KS; sb += 5) {
                    double a0[5], a1[5];
#pragma unroll
                    for (int u = 0; u < 5; ++u) {
                        const bool in = sb + u < KS && rb0 + 4 * (sb + u) < ld;
                        a0[u] = (in && !z0) ? g0[4 * (sb + u)] : 0.0;
                        if (NT > 1) a1[u] = (in && !z1) ? g1[4 * (sb + u)] : 0.0;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 5; ++u)
                        if (sb + u < KS) {
                            acc[0] = mfma_f64(a0[u], rf[sb + u], acc[0]);
                            if (NT > 1) acc[NT > 1 ? 1 : 0] = mfma_f64(a1[u], rf[sb + u], acc[NT > 1 ? 1 : 0]);
                        }
                }
            }
            // (register q of tile t holds the columns 16 t + 4 q + (lane >> 4): group 4 t + q)
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (4 * t + q < KF) part[((wave * KF + 4 * t + q)) * 64 + lane] = acc[t][q];
        }
        tick(11);
        __syncthreads();
        tick(12);
        // ---- one thread per (cell, chain): decision of a chain that finished, gradient, leapfrog update
        {
#pragma unroll
            for (int v = 0; v < RB_WAVES; ++v) ud += redu[v * 16 + lo4];
            const double tR = tots[lo4], tP1 = tots[16 + lo4], tP0 = tots[32 + lo4];
            double S = 0.0;
            if (jt < cpw4) {
#pragma unroll
                for (int v = 0; v < RB_WAVES; ++v) S += part[(v * KF + (jt >> 2)) * 64 + (jt & 3) * 16 + ct];
            }
            sh_r = sh_pp1 = sh_pp0 = 0.0;
            bool evaluate = (mode == RUN);
            tick(13);
            if (mode == DECIDE) {
                const double Unew = udL + a.alpha * tR;
                const double Hcur = 0.5 * pp0 + U0, Hnew = 0.5 * tP1 + Unew;
                const bool acc = (Hnew < Hcur) || (uc < exp(-(Hnew - Hcur)));  // hmc.py:158-177
                if (acc) {
                    U0 = Unew;
                    U1 = udL;
                    U2 = tR;
                    xc = xprop;
                    gc = gprop;
                }
                const int slot = ct * Tout + n_done;
                if (acc && a.xacc && cell) a.xacc[(int64_t)slot * M + jg] = xprop;
                if (w == 0 && tid < 16 && ct < C) {
                    a.accepted[slot] = acc ? 1 : 0;
                    double *o = a.out5s + 5 * slot;
                    o[0] = U0;
                    o[1] = U1;
                    o[2] = U2;
                    o[3] = Hcur;
                    o[4] = Hnew;
                }
                n_done += 1;
                if (spec) {
                    // the next trajectory's first step was taken from the proposal
                    Lc = Ln;
                    uc = un;
                    q_next += 1;
                    fetched = false;
                    if (Lc <= 2) fetch_next();  // (a trajectory this short may see its last evaluation in this very lock-step)
                    s_done = 0;
                    mode = RUN;
                    if (acc) {
                        evaluate = true;  // the evaluation that arrived is the new trajectory's first
                    } else {
                        // start again from the current sample: this lock-step of the chain is lost
                        if (cell) {
                            start_from(xc, gc, p0k);
                            sh_pp0 = p0k * p0k;
                        }
                        lost += 1;
                    }
                } else {
                    mode = IDLE;
                }
                spec = false;
            }
            tick(5);
            if (evaluate) {
                evals += 1;
                // regulariser gradient of the own cell at xs (potential.py:719-810)
                double val = 0.0, g = 0.0;
                if (cell) {
                    if (stencil) {
                        ra.x = a.xpub + ((int64_t)par * C + ct) * M;
                        g = a.alpha * reg_cell<false, true>(ra, jg, xs, val);
                    } else {
                        const double v = xs - apr;
                        if (a.kind == 0) {
                            val = v * v;
                            g = a.alpha * (2.0 * v);
                        } else {
                            const double v2 = v * v, den = v2 + a.beta;
                            val = (w2 * v2) / den;
                            g = a.alpha * ((2.0 * a.beta * w2 * v) / (den * den));
                        }
                    }
                    g = 2.0 * S + g;
                }
                if (s_done == 0) pp0 = tP0;  // (the sum of the trajectory's p'p before travelled with its first evaluation)
                s_done += 1;
                if (s_done < Lc) {
                    if (cell) {
                        double pj = ps - a.dt * g;
                        double xj = xs + a.dt * pj;
                        if (xj > chi) {
                            xj = chi;
                            pj = -pj;
                        } else if (xj < clo) {
                            xj = clo;
                            pj = -pj;
                        }
                        ps = pj;
                        xs = xj;
                    }
                } else {
                    // last evaluation: final half momentum step; the proposal and its gradient are kept
                    udL = ud;
                    mode = DECIDE;
                    spec = q_next < T;
                    if (cell) {
                        const double pf = ps - 0.5 * a.dt * g;
                        sh_pp1 = pf * pf;
                        sh_r = val;
                        xprop = xs;
                        gprop = g;
                        if (spec) {
                            p0k = p0n;
                            start_from(xprop, gprop, p0k);
                            sh_pp0 = p0k * p0k;
                        }
                    }
                }
            }
            tick(8);
            if (cell) Xs[jt * 16 + ct] = xs;
            post_shares();
        }
        tick(14);
    }
    // ---- state of every chain for the next launch
    __syncthreads();
    if (cell) {
        a.x_cur[(int64_t)ct * M + jg] = xc;
        a.g_cur[(int64_t)ct * M + jg] = gc;
        a.xs_io[(int64_t)ct * M + jg] = xs;
        a.ps_io[(int64_t)ct * M + jg] = ps;
        a.pst_io[(int64_t)ct * M + jg] = p0k;
    }
    if (w == 0 && tid < 16 && ct < C) {
        a.u_cur[3 * ct] = U0;
        a.u_cur[3 * ct + 1] = U1;
        a.u_cur[3 * ct + 2] = U2;
        double *cs = a.cst_io + RB_CST * ct;
        cs[0] = (mode == RUN) ? 1.0 : 0.0;
        cs[1] = (double)s_done;
        cs[2] = (double)Lc;
        cs[3] = uc;
        cs[4] = pp0;
        a.n_io[ct] = q_next;
        a.n_io[16 + ct] = n_done;
        a.n_io[48 + ct] = lost;
        a.n_io[64 + ct] = (mode == RUN) ? 1 : 0;
        a.n_io[80 + ct] = evals;
        if (tid == 0) a.n_io[32] = step;
    }
    if (timing) {
        tick(10);
        for (int i = 0; i < 16; ++i) a.dbg[(w == 0 ? 0 : 16) + i] += tacc_s[i];
    }
}

}  // namespace ghk
