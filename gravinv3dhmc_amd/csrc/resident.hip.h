// Resident chain kernel (gfx950): a whole batch of HMC trajectories in ONE launch for sensitivity
// matrices small enough to live on the chip (256 CUs x (160 KB LDS + registers): N <= 1024 and
// N*M*8 up to ~36 MB with a copy in LDS, ~65 MB with one copy split between LDS and registers --
// the reference's uniformgrid and realdata examples, BASELINE configs[0] (600 x 6000) and, through
// the dense form of the compressed forward, configs[2]).
//
// At that size the sweep-per-launch path is bound by launches and by re-reading G from L2/MALL
// (3 launches, ~30 us per leapfrog step at C1), not by HBM.  Here every workgroup owns a fixed
// block of columns (cells), loads them once per launch and keeps them; per leapfrog step only
// N-vectors cross the chip:
//
//   local        wave-per-column dots <G_j, r> (columns in registers where they fit) -> gradient ->
//                one thread per column: momentum/position update with clamp-and-reflect
//                (hmc.py:114-152); thread-per-row forward partial sum_j G_ij x_j over the own columns
//   level 1      cluster = workgroups with equal w % 8 (one XCD, one L2).  The partials are
//                published as tagged granules; member `crank` of every cluster sums row chunk
//                `crank` of d over its cluster (through L2 when the placement allows it)
//   level 2      the chunk's eight cluster sums cross the XCDs once (write-through); every cluster's
//                owner adds them in the same order -> the same d everywhere
//   level 3      every workgroup reads d from its cluster's copy, removes the mean and forms r
//                redundantly (potential.py:700-706: identical bits everywhere); regulariser gradient
//                of its own cells (Smoothness / TV: one thread per (cell, neighbour), the neighbour
//                requested from the published model as soon as d is in, the terms formed by the
//                next dots pass)
//   trajectory   last half momentum step; three scalars per workgroup (R, p'p before/after) summed
//   end          over the cluster by its member 0, the eight cluster sums read by everybody;
//                Metropolis test (hmc.py:158-177) decided identically by every workgroup; the
//                gradient at the proposal is kept for the next trajectory, so a trajectory of L
//                steps costs exactly L evaluations
//
// The kernel is bound by latency (SQ counters in profiles/r01: VALU 6 %, LDS 3 % busy, 1 % bank
// conflicts), so what matters inside a workgroup is how many dependent round trips a step makes:
// LDS and global reads are issued in batches in front of scheduling fences
// (__builtin_amdgcn_sched_barrier) -- at 256 VGPRs the scheduler otherwise orders read, wait, use
// one by one.
//
// Inter-workgroup hand-offs: the data is the flag.  A double travels as two naturally aligned
// 8-byte granules {tag = evaluation number, 32 bits of the value}, each written by ONE store and
// read by sc1 loads that bypass the reader's L1; a reader re-reads the granules it still misses
// until every tag is the evaluation it waits for.  No counters, no fences, no grid barrier; a
// buffer is only overwritten after every reader has published something that depends on having
// read it (the cross-XCD sums are double-buffered by evaluation parity).  Measured steps from the
// first version (arrival counters + agent-scope acquires, 19.9 us per evaluation at C1) to this
// one (6.4 us): DESIGN.md 4.5.  Every spin is bounded: on a time-out the abort word is raised,
// every workgroup leaves and the host falls back to the sweep path.  The grid (one workgroup per
// CU by the LDS request) is checked against the occupancy query on the host.
#pragma once
#include "kernels.hip.h"

namespace ghk {

constexpr int RES_THREADS = 512;
constexpr int RES_WAVES = 8;
constexpr int RES_CLUSTERS = 8;                       // logical clusters: workgroups w with equal w % 8
constexpr int RES_CHUNKS = 32;                        // at most: row chunks of d, one per owning member of a cluster
constexpr int RES_MAX_WG = 256;                       // <= 32 members per cluster (two per reducing thread)
constexpr int RES_REDBUF = 16 * 33;                   // LDS transpose buffer of the cluster reduction
constexpr long long RES_TIMEOUT_TICKS = 200000000LL;  // 2 s of the 100 MHz wall clock, per wait

using u64 = unsigned long long;

struct ResArgs {
    const double *G;   // Aw: the dots pass (adjoint)
    const double *Gl;  // the forward pass' operator, held in LDS: Aw itself, or the dense model-space
                       // form F = Awcp W of the wavelet-compressed forward (then CW > 0: the dots
                       // read Aw from their register copy)
    int64_t ld, N, M;
    int cols_per_wg, nwg;
    int split;      // 1 (CW > 0, Gl == G): ONE copy of the workgroup's columns, the first 8 CW of them in
                    // the waves' registers, only the rest in LDS -- for kernels up to ~1.8x the LDS
    int stream;     // 1 (CW == 0): columns [lds_cap, cols_per_wg) of the workgroup are not resident: both
    int lds_cap;    // passes read them from global memory (L2 / Infinity Cache) in every evaluation --
                    // kernels of up to a few hundred MB with N <= 1024 (the reference's ratiogrid
                    // example: 123 MB + its dense compressed-forward form).  With Gl != G (wavelet
                    // forward) the dots read ALL their columns of Aw from memory.
    int try_local;  // 1: keep intra-cluster traffic in the XCD's L2 when the placement allows it
    const double *gfix, *dobs_c, *low, *high;
    // regulariser (x is set per evaluation inside the kernel)
    int kind, nz, ny, nx;
    double alpha, beta;
    const double *mwapr, *wm2;
    // chains: C of them share the launch (and the resident G); trajectory k belongs to chain
    // chain[k] (nullptr: chain 0) and the chains are advanced in the order of the list
    int C;
    const int *chain;
    double *x_cur;     // C x M: in = current models, out = models after the last trajectories run
    double *gcur_io;   // C x M full gradient at the current models (kept between launches), or nullptr
    double *ucur_io;   // 3 C: {U, U_data, R} of the current models
    int have_state;    // 1: gcur_io / ucur_io are valid for x_cur (skip the evaluation at launch)
    int K;
    const int *L;
    const double *p0s;  // K x M momenta, drawn by the host in the reference's order
    const double *us;   // K uniforms of the Metropolis test
    double dt;
    long long stop_at_accepts, accept_count0;
    // outputs
    int *accepted;   // K
    double *out5s;   // K x {U, U_data, R, H_current, H_proposal}
    double *xacc;    // K x M accepted models (nullptr: not wanted)
    int *n_run;      // [0] trajectories run, [1] evaluations, [2] scalar gathers
    // workspace (granule buffers keep their tags across launches: tag0 / tagE0 continue the count)
    u64 *slabg;      // nwg x ld x 2 forward partials
    u64 *xslabg;     // 2 x 8 x ld x 2 cluster sums, double-buffered by evaluation parity
    u64 *dclg;       // 8 x ld x 2 finished d, one copy per cluster
    u64 *xccg;       // nwg: {launch tag, XCC id} of every workgroup
    u64 *scalg;      // nwg x 8 trajectory-end scalars of the workgroups
    u64 *xscalg;     // 8 x 8 the same summed over each cluster
    double *xpub;    // 2 x M models as the stencil regularisers see them (Smoothness / TV)
    unsigned tag0, tagE0;
    unsigned *abort_w;
    long long *dbg;  // optional: 2 x 16 accumulated phase times (100 MHz ticks) of the first / last workgroup
};

constexpr int RES_MAX_CHAINS = 16;

// lds_cols: columns of the workgroup held in LDS (all of them, or those beyond the register-held
// ones in split mode, which also needs 8 x ld doubles of scratch for the waves' forward partials)
static inline size_t resident_lds_doubles(int64_t ld, int cols_per_wg, int chains, int lds_cols, bool split)
{
    return (size_t)lds_cols * (size_t)ld + (split ? 8 * (size_t)ld : 0) + (size_t)ld + RES_REDBUF + 16 +
           (6 + 2 * (size_t)chains) * (size_t)cols_per_wg + 3 * RES_MAX_CHAINS + 8 + 16;
}

// The stencil regularisers (Smoothness, TV) take one thread per (cell, neighbour) and 12 doubles of
// the reduction buffer per cell: shapes with more cells per workgroup stay on the sweep path.
static inline bool resident_stencil_fits(int cols_per_wg)
{
    return 6 * cols_per_wg <= RES_THREADS && 12 * cols_per_wg <= RES_REDBUF;
}

// 8-byte write-through store (global_store_dwordx2 sc1)
__device__ __forceinline__ void st_wt(double *p, double v)
{
    __hip_atomic_store(reinterpret_cast<u64 *>(p), (u64)__double_as_longlong(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);
}

// one double as two tagged granules at g[0], g[1]
__device__ __forceinline__ void st_gran(u64 *g, unsigned tag, double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    __hip_atomic_store(g, ((u64)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(g + 1, ((u64)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the same with plain stores: the line stays in the XCD's L2, where sc1 loads of the same XCD's
// CUs find it (NOT visible to other XCDs until it is written back)
__device__ __forceinline__ void st_gran_l2(u64 *g, unsigned tag, double v)
{
    const u64 b = (u64)__double_as_longlong(v);
    __hip_atomic_store(g, ((u64)tag << 32) | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    __hip_atomic_store(g + 1, ((u64)tag << 32) | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// the two halves of ld_gran: the loads (issue early), and what they brought (look late -- the wait
// for the loads sits where the words are first used)
__device__ __forceinline__ void ld_gran_issue(u64 *g, u64 &a, u64 &b)
{
    a = __hip_atomic_load(g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    b = __hip_atomic_load(g + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ bool gran_value(u64 a, u64 b, unsigned tag, double &v)
{
    v = __longlong_as_double((long long)((a & 0xffffffffull) | (b << 32)));
    return (unsigned)(a >> 32) == tag && (unsigned)(b >> 32) == tag;
}

__device__ __forceinline__ bool ld_gran(u64 *g, unsigned tag, double &v)
{
    u64 a, b;
    ld_gran_issue(g, a, b);
    return gran_value(a, b, tag, v);
}

// Every lane of the wave re-reads its granules (try_load: true when all of them carry the tag)
// until the whole wave has them.  false: timed out or another workgroup raised the abort word.
template <typename F>
__device__ __forceinline__ bool res_poll(unsigned *abort_w, F &&try_load)
{
    unsigned spins = 0;
    long long t0 = 0;
    for (;;) {
        const bool ok = try_load();
        if (__all(ok)) return true;
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 63u) == 0) {
            const long long now = wall_clock64();
            if (t0 == 0) t0 = now;
            if (__hip_atomic_load(abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                now - t0 > RES_TIMEOUT_TICKS) {
                __hip_atomic_store(abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return false;
            }
        }
    }
}

// Workgroup-wide sum (RES_WAVES waves), fixed order, result valid in every thread
__device__ __forceinline__ double res_block_sum(double v, double *red)
{
    v = wave_sum_dpp(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    // (the eight partials in four 16-byte reads issued together; summed in wave order)
    const d2 *red2 = reinterpret_cast<const d2 *>(red);
    const d2 a0 = red2[0], a1 = red2[1], a2 = red2[2], a3 = red2[3];
    __builtin_amdgcn_sched_barrier(0);
    double t = 0.0;
    t += a0.x;
    t += a0.y;
    t += a1.x;
    t += a1.y;
    t += a2.x;
    t += a2.y;
    t += a3.x;
    t += a3.y;
    return t;
}

// RC = double2 chunks a lane holds of one column / of r: 64*RC*2 >= ld.
// CW > 0: every wave keeps its (at most CW = ceil(columns per workgroup / 8)) columns of the
// wave-per-column dots pass in registers (4 CW RC VGPRs) as well: LDS bandwidth bounds the local
// part of a step, and with this copy the LDS one is read once per step (by the forward pass)
// instead of twice.  Measured at C1 (RC 5, CW 3): 8.3 -> 7.1 us per evaluation; a register copy
// for the forward pass instead (thread = row pair, 96 VGPRs) gave 7.7, both together spill.
template <int RC, int CW>
__global__ void __launch_bounds__(RES_THREADS) resident_chain_kernel(ResArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int w = blockIdx.x;
    const int ld = (int)a.ld, ld2 = ld >> 1;
    const int cpw = a.cols_per_wg;
    const int64_t M = a.M;
    const int64_t j0 = (int64_t)w * cpw;
    const int nc = (int)((M - j0 < cpw) ? (M - j0) : cpw);
    const int nwg = a.nwg;
    const bool stencil = (a.kind == 1 || a.kind == 3);
    // (stencil regularisers: one thread per (cell, neighbour), resident_stencil_fits)

    // split mode: columns [0, 8 CW) of the workgroup live in the waves' registers only
    const bool split = CW > 0 && a.split != 0;
    const bool stream = CW == 0 && a.stream != 0;
    const int lds_c0 = split ? CW * RES_WAVES : 0;             // first column held in LDS
    const int lds_cols = stream ? a.lds_cap : (cpw > lds_c0 ? cpw - lds_c0 : 0);  // capacity (columns)
    const int nl = stream ? (nc < a.lds_cap ? nc : a.lds_cap)
                          : (nc > lds_c0 ? nc - lds_c0 : 0);   // LDS columns of this workgroup
    double *Gs = smem;                                   // lds_cols x ld
    double *fsc = Gs + (size_t)lds_cols * ld;            // split: 8 x ld forward partials of the waves
    double *r_s = fsc + (split ? 8 * (size_t)ld : 0);    // ld
    double *redbuf = r_s + ld;                 // 16 x 33
    double *red = redbuf + RES_REDBUF;         // 16
    double *xs = red + 16;                     // position of the running trajectory
    double *ps = xs + cpw;                     // momentum
    double *gs = ps + cpw;                     // full gradient at the latest evaluation
    double *gr = gs + cpw;                     // alpha * dR/dx at the latest evaluation
    double *lo = gr + cpw;
    double *hi = lo + cpw;
    double *xc_all = hi + cpw;                 // C x cpw: current sample of every chain
    double *gc_all = xc_all + (size_t)a.C * cpw;   // C x cpw: full gradient there
    double *ucs = gc_all + (size_t)a.C * cpw;  // 3 x RES_MAX_CHAINS: {U, U_data, R} of the current samples
    int *flag_s = reinterpret_cast<int *>(ucs + 3 * RES_MAX_CHAINS);  // 1 while no wave gave up
    const d2 *Gs2 = reinterpret_cast<const d2 *>(Gs);
    d2 *r_s2 = reinterpret_cast<d2 *>(r_s);

    int ev = 0, ng = 0;  // evaluations / scalar gathers so far in this launch
    // per-phase clocks of thread 0 (diagnostic, GRAVHMC_RESIDENT_TIMING): accumulated in LDS -- as
    // registers they would cost every thread 32 VGPRs in a kernel that has none to spare
    long long *tacc_s = reinterpret_cast<long long *>(ucs + 3 * RES_MAX_CHAINS + 8);
    long long tlast = 0;
    const bool timing = a.dbg != nullptr && tid == 0 && (w == 0 || w == a.nwg - 1);
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

    // ---- load the workgroup's columns (contiguous in the column-major G) and per-cell vectors
    {
        const d2 *src = reinterpret_cast<const d2 *>(a.Gl + (j0 + lds_c0) * a.ld);
        d2 *dst = reinterpret_cast<d2 *>(Gs);
        const int tot = nl * ld2;
        for (int e = tid; e < tot; e += RES_THREADS) dst[e] = __builtin_nontemporal_load(src + e);
    }
    if (tid < nc) {
        const int64_t j = j0 + tid;
        for (int c = 0; c < a.C; ++c) {
            xc_all[c * cpw + tid] = a.x_cur[(int64_t)c * M + j];
            if (a.have_state) gc_all[c * cpw + tid] = a.gcur_io[(int64_t)c * M + j];
        }
        xs[tid] = xc_all[tid];
        ps[tid] = 0.0;
        lo[tid] = a.low[j];
        hi[tid] = a.high[j];
    }
    if (a.have_state && tid < 3 * a.C) ucs[tid] = a.ucur_io[tid];
    if (tid == 0) *flag_s = 1;
    d2 gq_reg[CW > 0 ? CW : 1][RC];
    if (CW > 0) {
#pragma unroll
        for (int q = 0; q < CW; ++q) {
            const int c = wave + q * RES_WAVES;
            const d2 *col = reinterpret_cast<const d2 *>(a.G + (j0 + c) * a.ld);
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                const int e = lane + 64 * k;
                gq_reg[q][k] = (c < nc && e < ld2) ? col[e] : d2{0.0, 0.0};
            }
        }
    }

    // Logical clusters: workgroups with equal w % 8 (the dispatcher deals workgroups round-robin
    // over the 8 XCDs, so a cluster normally shares one L2).  The reduction order is defined by the
    // logical indices only; the placement decides how cluster-internal data is published: plain
    // stores that stay in the shared L2 when every member reports the same XCC id, write-through
    // stores otherwise (correct under any placement).
    const int ncl = nwg < RES_CLUSTERS ? nwg : RES_CLUSTERS;
    const int cg = w % RES_CLUSTERS, crank = w / RES_CLUSTERS;
    const int cn = (nwg - cg + RES_CLUSTERS - 1) / RES_CLUSTERS;  // members of this cluster
    // d is cut in nch row chunks; member `crank` of every cluster owns chunk `crank` (nch = size of
    // the smallest cluster, so nobody owns two: the chunks of an evaluation proceed in parallel)
    const int nch = (nwg / RES_CLUSTERS) < 1 ? 1 : ((nwg / RES_CLUSTERS) > RES_CHUNKS ? RES_CHUNKS : nwg / RES_CLUSTERS);
    const int ch = (ld + nch - 1) / nch;  // rows per chunk
    bool local = false;
    {
        const unsigned ltag = a.tagE0 + 1u;  // unique per launch
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID[3:0]
        if (tid == 0)
            __hip_atomic_store(a.xccg + w, ((u64)ltag << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (wave == 0) {
            bool same = true;
            const bool got = res_poll(a.abort_w, [&]() -> bool {
                bool ok = true;
                if (lane < cn) {
                    const u64 e = __hip_atomic_load(a.xccg + cg + RES_CLUSTERS * lane, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
                    ok = (unsigned)(e >> 32) == ltag;
                    same = ((unsigned)e & 0xfu) == xcc;
                }
                return ok;
            });
            if (lane == 0) red[0] = (got && __all(same) && a.try_local) ? 1.0 : 0.0;
            if (!got) *flag_s = 0;
        }
        __syncthreads();
        local = red[0] != 0.0;
        __syncthreads();
        if (*flag_s == 0) return;
    }
    // rows 2 tid, 2 tid + 1 of the constant data vectors live in registers
    const int i0 = 2 * tid;
    d2 gf = d2{0.0, 0.0}, dc = d2{0.0, 0.0};
    if (tid < ld2) {
        if (i0 < a.N) {
            dc.x = a.dobs_c[i0];
            if (a.gfix) gf.x = a.gfix[i0];
        }
        if (i0 + 1 < a.N) {
            dc.y = a.dobs_c[i0 + 1];
            if (a.gfix) gf.y = a.gfix[i0 + 1];
        }
    }

    // prior model / MS weight of the own cell (thread tid < nc)
    double apr_j = 0.0, w2_j = 1.0;
    if (tid < nc) {
        apr_j = a.mwapr[j0 + tid];
        if (a.kind == 2) w2_j = a.wm2[j0 + tid];
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
    ra.x = nullptr;
    ra.greg = nullptr;
    ra.regpart = nullptr;

    // Forward product of xs through both hops: on return r_s holds the residual and gr the
    // regulariser gradient; with `last` also ud (data misfit) and Rw (this workgroup's share of
    // R) -- only the end of a trajectory needs the potential itself.  false: aborted.
    double ud = 0.0, Rw = 0.0;
    // stencil regularisers, one thread per (cell, neighbour): the neighbour's model and prior model,
    // the cell's prior model, whether the neighbour exists; gr_pending: the terms of the latest
    // evaluation are still to be formed from them (by the dots pass, right before it needs gr:
    // the residual and the dots in between hide the latency of these loads)
    double xn1 = 0.0, an1 = 0.0, ap1 = 0.0;
    bool ex1 = false, gr_pending = false;
    const int c6 = tid / 6, q6 = tid - 6 * c6;
    int nb_off = 0;  // the neighbour's cell index minus the cell's
    if (stencil && c6 < nc) {
        const int64_t j = j0 + c6, P = (int64_t)a.nx * a.ny;
        const int64_t ci = j % a.nx, cj = (j / a.nx) % a.ny, ck = j / P;
        const int ax = q6 >> 1;
        const int64_t pos = ax == 0 ? ci : ax == 1 ? cj : ck, len = ax == 0 ? a.nx : ax == 1 ? a.ny : a.nz;
        const int64_t str = ax == 0 ? 1 : ax == 1 ? a.nx : P;
        ex1 = (q6 & 1) ? pos > 0 : pos < len - 1;
        nb_off = ex1 ? (int)((q6 & 1) ? -str : str) : 0;
        an1 = a.mwapr[j + nb_off];
        ap1 = a.mwapr[j];
    }
    // term q6 of cell c6 -> redbuf[tid] (dR/dx share, signed) and, with_val, redbuf[6 cpw + tid] (R share)
    auto stencil_terms = [&](bool with_val) {
        if (c6 < nc) {
            const double v = xs[c6] - ap1;
            const double t = (q6 & 1) ? (xn1 - an1) - v : v - (xn1 - an1);
            double gt, vt;
            if (a.kind == 1) {  // Smoothness (potential.py:786-796)
                gt = 2.0 * t;
                vt = t * t;
            } else {  // TV (potential.py:798-810)
                vt = sqrt(t * t + a.beta);
                gt = t / vt;
            }
            redbuf[tid] = ex1 ? ((q6 & 1) ? -gt : gt) : 0.0;
            if (with_val) redbuf[6 * cpw + tid] = (ex1 && !(q6 & 1)) ? vt : 0.0;
        }
    };
    auto evaluate = [&](bool last) -> bool {
        __syncthreads();  // xs complete
        tick(0);
        const unsigned tag = a.tag0 + (unsigned)ev + 1u;
        const int par = ev & 1;
        // neighbours read the model once they have d (level 3): these stores are drained (below,
        // behind the forward pass that hides their latency) before any of this workgroup's
        // partials, which d transitively depends on, is published
        if (stencil && tid < nc) st_wt(a.xpub + (int64_t)par * M + j0 + tid, xs[tid]);
        if (split) {
            // forward share of the register-held columns: per wave over its columns, then summed
            // over the waves through LDS in a fixed order
            d2 dacc[RC];
#pragma unroll
            for (int k = 0; k < RC; ++k) dacc[k] = d2{0.0, 0.0};
#pragma unroll
            for (int q = 0; q < (CW > 0 ? CW : 1); ++q) {
                const int c = wave + q * RES_WAVES;
                if (c < nc) {
                    const double x = xs[c];
#pragma unroll
                    for (int k = 0; k < RC; ++k) {
                        dacc[k].x += gq_reg[q][k].x * x;
                        dacc[k].y += gq_reg[q][k].y * x;
                    }
                }
            }
            d2 *f2 = reinterpret_cast<d2 *>(fsc) + wave * ld2;
#pragma unroll
            for (int k = 0; k < RC; ++k) {
                const int e = lane + 64 * k;
                if (e < ld2) f2[e] = dacc[k];
            }
            __syncthreads();
        }
        d2 facc = d2{0.0, 0.0};
        if (tid < ld2) {
            // (LDS reads in batches of eight with the arithmetic fenced off behind them: at this
            // kernel's register pressure the scheduler otherwise issues read, wait, FMA one by one
            // and the pass runs at LDS latency -- measured 1.4 us for 24 columns -- instead of
            // LDS bandwidth; the order of the sums is unchanged)
            if (split) {
                const d2 *f2 = reinterpret_cast<const d2 *>(fsc);
                d2 t[RES_WAVES];
#pragma unroll
                for (int v = 0; v < RES_WAVES; ++v) t[v] = f2[v * ld2 + tid];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int v = 0; v < RES_WAVES; ++v) {
                    facc.x += t[v].x;
                    facc.y += t[v].y;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            int c = 0;
            for (; c + 8 <= nl; c += 8) {
                d2 g[8];
                double x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) g[u] = Gs2[(c + u) * ld2 + tid];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = xs[lds_c0 + c + u];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    facc.x += g[u].x * x[u];
                    facc.y += g[u].y * x[u];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            for (; c < nl; ++c) {
                const d2 g = Gs2[c * ld2 + tid];
                const double x = xs[lds_c0 + c];
                facc.x += g.x * x;
                facc.y += g.y * x;
            }
            if (stream) {
                // the columns that are not resident: coalesced reads of the forward operator from
                // L2 / Infinity Cache, eight in flight per thread, same order of the sums
                const d2 *Gg = reinterpret_cast<const d2 *>(a.Gl + j0 * a.ld);
                for (c = nl; c + 8 <= nc; c += 8) {
                    d2 g[8];
                    double x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) g[u] = Gg[(int64_t)(c + u) * ld2 + tid];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = xs[c + u];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        facc.x += g[u].x * x[u];
                        facc.y += g[u].y * x[u];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                for (; c < nc; ++c) {
                    const d2 g = Gg[(int64_t)c * ld2 + tid];
                    const double x = xs[c];
                    facc.x += g.x * x;
                    facc.y += g.y * x;
                }
            }
        }
        if (stencil) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (tid < ld2) {
            u64 *out = a.slabg + ((int64_t)w * ld + i0) * 2;
            if (local) {
                st_gran_l2(out, tag, facc.x);
                st_gran_l2(out + 2, tag, facc.y);
            } else {
                st_gran(out, tag, facc.x);
                st_gran(out + 2, tag, facc.y);
            }
        }
        // the cell-local regularisers (Damping, MS) do not wait for anybody
        double val = 0.0;
        if (!stencil && tid < nc) {
            const double v = xs[tid] - apr_j;
            if (a.kind == 0) {  // Damping (potential.py:719-723)
                val = v * v;
                gr[tid] = a.alpha * (2.0 * v);
            } else {  // MS (potential.py:775-788)
                const double v2 = v * v, den = v2 + a.beta;
                val = (w2_j * v2) / den;
                gr[tid] = a.alpha * ((2.0 * a.beta * w2_j * v) / (den * den));
            }
        }
        tick(1);
        // level 1 + 2: the owner of row chunk `crank` sums it over the cluster, then over the clusters
        if (crank < nch) {
            const int k = crank;
            for (int rb = 0; rb < ch; rb += 32) {  // 32 rows per pass
                {
                    // cluster sum: thread = (member group mg, row); members mg, mg + 16
                    const int row = rb + (tid & 31), mg = tid >> 5;
                    const int i = k * ch + row;
                    const bool act = row < ch && i < ld;
                    double v0 = 0.0, v1 = 0.0;
                    bool h0 = !(act && mg < cn), h1 = !(act && mg + 16 < cn);
                    const bool got = res_poll(a.abort_w, [&]() -> bool {
                        if (!h0) h0 = ld_gran(a.slabg + ((int64_t)(cg + RES_CLUSTERS * mg) * ld + i) * 2, tag, v0);
                        if (!h1)
                            h1 = ld_gran(a.slabg + ((int64_t)(cg + RES_CLUSTERS * (mg + 16)) * ld + i) * 2, tag, v1);
                        return h0 && h1;
                    });
                    if (!got) *flag_s = 0;
                    redbuf[mg * 33 + (tid & 31)] = v0 + v1;
                }
                __syncthreads();
                tick(3);
                // transposed: 16 lanes = the 16 member groups of one row, summed on the VALU
                const int row = rb + (tid >> 4), sub = tid & 15;
                const int i = k * ch + row;
                const bool act = row < ch && i < ld;
                const double csum = row16_sum_dpp(redbuf[sub * 33 + (tid >> 4)]);
                // (parity buffers: a slow owner of this chunk in another cluster may still be reading
                // the previous evaluation's sums when this cluster is already one evaluation ahead)
                u64 *xs_e = a.xslabg + (int64_t)par * RES_CLUSTERS * ld * 2;
                if (act && sub == 15) st_gran(xs_e + ((int64_t)cg * ld + i) * 2, tag, csum);
                // sum over the clusters, read back from all of them (own one included): every
                // cluster computes the same bits
                double u = 0.0;
                bool hu = !(act && sub < ncl);
                const bool got = res_poll(a.abort_w, [&]() -> bool {
                    if (!hu) hu = ld_gran(xs_e + ((int64_t)sub * ld + i) * 2, tag, u);
                    return hu;
                });
                if (!got) *flag_s = 0;
                const double dtot = row16_sum_dpp(u);
                if (act && sub == 15) {
                    u64 *out = a.dclg + ((int64_t)cg * ld + i) * 2;
                    if (local)
                        st_gran_l2(out, tag, dtot);
                    else
                        st_gran(out, tag, dtot);
                }
                __syncthreads();  // redbuf is reused by the next pass
                tick(4);
            }
        }
        // level 3: the finished d from this cluster's copy
        d2 dinv = d2{0.0, 0.0};
        {
            double dx = 0.0, dy = 0.0;
            bool hx = !(tid < ld2), hy = hx;
            u64 *in = a.dclg + ((int64_t)cg * ld + i0) * 2;
            const bool got = res_poll(a.abort_w, [&]() -> bool {
                if (!hx) hx = ld_gran(in, tag, dx);
                if (!hy) hy = ld_gran(in + 2, tag, dy);
                return hx && hy;
            });
            if (!got) *flag_s = 0;
            dinv = d2{dx, dy};
        }
        tick(6);
        // Stencil regularisers: the neighbours' models are requested now and used behind the
        // residual below, which hides the memory latency.  Any row of d depends on a partial of
        // every workgroup, and those were published after the workgroup's model stores had drained
        // (write-through); the loads bypass L1: no fence needed.
        // Thread 6 c + q takes term q (axis
        // q / 2, forward / backward) of cell c -- one square root and one division per lane
        // instead of six of each on a few lanes of wave 0 -- and the terms meet again in LDS, added
        // in the order reg_stencil_eval adds them.
        if (stencil) ra.x = a.xpub + (int64_t)par * M;
        if (stencil) {
            if (c6 < nc)
                xn1 = __longlong_as_double((long long)__hip_atomic_load(
                    reinterpret_cast<const unsigned long long *>(ra.x + (j0 + c6 + nb_off)), __ATOMIC_RELAXED,
                    __HIP_MEMORY_SCOPE_AGENT));
            __builtin_amdgcn_sched_barrier(0);
        }
        // mean removal, residual, data misfit (potential.py:700-706)
        double s = 0.0;
        if (tid < ld2) {
            dinv.x += gf.x;
            dinv.y += gf.y;
            if (i0 < a.N) s += dinv.x;
            if (i0 + 1 < a.N) s += dinv.y;
        }
        const double mean = res_block_sum(s, red) / (double)a.N;
        if (*flag_s == 0) return false;  // (behind the barriers of the reduction: uniform)
        double acc = 0.0;
        if (tid < ld2) {
            d2 rv = d2{0.0, 0.0};
            if (i0 < a.N) {
                rv.x = (dinv.x - mean) - dc.x;
                acc += rv.x * rv.x;
            }
            if (i0 + 1 < a.N) {
                rv.y = (dinv.y - mean) - dc.y;
                acc += rv.y * rv.y;
            }
            r_s2[tid] = rv;
        }
        // regulariser of the own cells at xs, neighbours from the published model (requested above)
        if (stencil && !last) {
            gr_pending = true;  // formed by the dots pass that follows
        } else if (stencil) {
            stencil_terms(true);
            __syncthreads();
            if (tid < nc) {
                double g = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) g += redbuf[6 * tid + q];
#pragma unroll
                for (int q = 0; q < 6; q += 2) val += redbuf[6 * cpw + 6 * tid + q];
                gr[tid] = a.alpha * g;
            }
        }
        if (last) {
            ud = res_block_sum(acc, red);
            Rw = res_block_sum(val, red);
        }
        ++ev;
        tick(7);
        return true;
    };

    // Wave-per-column dots with r_s (four columns of a wave in flight), then one thread per column:
    // full gradient and, what = 0: store it in gs; 1: leapfrog update with momentum coefficient
    // cu; 2: last half momentum step (returns this workgroup's sum of p^2, gradient in gs).
    auto dots = [&](int what, double cu) -> double {
        __syncthreads();  // r_s, gr complete
        tick(11);
        // (unconditional reads -- the few doubles past r_s belong to redbuf -- then zeroed: selects
        // instead of branches, all reads in flight together)
        d2 rr[RC];
#pragma unroll
        for (int k = 0; k < RC; ++k) rr[k] = r_s2[lane + 64 * k];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < RC; ++k)
            if (lane + 64 * k >= ld2) rr[k] = d2{0.0, 0.0};
        if (CW > 0) {
            double s[CW > 0 ? CW : 1];
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                s[q] = 0.0;
#pragma unroll
                for (int k = 0; k < RC; ++k) {
                    s[q] += gq_reg[q][k].x * rr[k].x;
                    s[q] += gq_reg[q][k].y * rr[k].y;
                }
            }
#pragma unroll
            for (int q = 0; q < CW; ++q) s[q] = wave_sum_dpp(s[q]);
#pragma unroll
            for (int q = 0; q < CW; ++q) {
                const int c = wave + q * RES_WAVES;
                if (lane == 0 && c < nc) gs[c] = s[q];
            }
        }
        if (CW == 0 || split) {
            // the columns held in LDS only
            for (int cb = lds_c0 + wave; cb < nc; cb += 4 * RES_WAVES) {
                double s[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = cb + q * RES_WAVES;
                    s[q] = 0.0;
                    if (c < nc) {
                        // (rows past the column read its last chunk again and meet rr = 0)
                        d2 g[RC];
                        // (stream mode: a column that is not resident -- or, with the compressed
                        // forward in LDS, any column of Aw -- comes from memory)
                        const d2 *colp = (stream && (c >= nl || a.G != a.Gl))
                                             ? reinterpret_cast<const d2 *>(a.G + (j0 + c) * a.ld)
                                             : Gs2 + (c - lds_c0) * ld2;
#pragma unroll
                        for (int k = 0; k < RC; ++k) {
                            const int e = lane + 64 * k;
                            g[k] = colp[e < ld2 ? e : ld2 - 1];
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int k = 0; k < RC; ++k) {
                            s[q] += g[k].x * rr[k].x;
                            s[q] += g[k].y * rr[k].y;
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] = wave_sum_dpp(s[q]);
                if (lane < 4) {
                    const int c = cb + lane * RES_WAVES;
                    const double sv = lane == 0 ? s[0] : lane == 1 ? s[1] : lane == 2 ? s[2] : s[3];
                    if (c < nc) gs[c] = sv;
                }
            }
        }
        if (gr_pending) stencil_terms(false);
        tick(12);
        __syncthreads();
        tick(13);
        double pp = 0.0;
        if (tid < nc) {
            double gr_j = gr[tid];
            if (gr_pending) {
                double g6 = 0.0;
#pragma unroll
                for (int q = 0; q < 6; ++q) g6 += redbuf[6 * tid + q];
                gr_j = a.alpha * g6;
                gr[tid] = gr_j;
            }
            const double gs_j = gs[tid], ps_j = ps[tid], xs_j = xs[tid];
            const double chi = hi[tid], clo = lo[tid];
            __builtin_amdgcn_sched_barrier(0);
            const double g = 2.0 * gs_j + gr_j;
            if (what == 1) {
                double pj = ps_j - cu * g;
                double xj = xs_j + a.dt * pj;
                if (xj > chi) {
                    xj = chi;
                    pj = -pj;
                } else if (xj < clo) {
                    xj = clo;
                    pj = -pj;
                }
                ps[tid] = pj;
                xs[tid] = xj;
            } else {
                if (what == 2) {
                    const double pf = ps_j - cu * g;
                    pp = pf * pf;
                }
                gs[tid] = g;
            }
        }
        gr_pending = false;
        tick(14);
        if (what != 2) return 0.0;
        return res_block_sum(pp, red);
    };

    // all-gather of {R share, p'p after, p'p before} over the workgroups, summed in a fixed order
    double Rtot = 0.0, pp1 = 0.0, pp0 = 0.0;
    auto gather_scalars = [&](double v_r, double v_pp1, double v_pp0) -> bool {
        // Two hops along the hardware like the N-vectors: the cluster's member 0 collects the
        // triples of its members (through L2 where the cluster shares one) and publishes the
        // cluster's sums; every workgroup then reads the eight cluster sums.  (One hop, every
        // workgroup reading all 250 triples from memory, measured 7 us per trajectory.)
        tick(8);
        const unsigned tag = a.tagE0 + (unsigned)ng + 1u;
        if (tid == 0) {
            u64 *out = a.scalg + 8 * w;
            if (local) {
                st_gran_l2(out, tag, v_r);
                st_gran_l2(out + 2, tag, v_pp1);
                st_gran_l2(out + 4, tag, v_pp0);
            } else {
                st_gran(out, tag, v_r);
                st_gran(out + 2, tag, v_pp1);
                st_gran(out + 4, tag, v_pp0);
            }
        }
        if (wave == 0) {
            if (crank == 0) {
                double t0 = 0.0, t1 = 0.0, t2 = 0.0;
                bool h0 = !(lane < cn), h1 = h0, h2 = h0;
                u64 *in = a.scalg + 8 * (cg + RES_CLUSTERS * (lane < cn ? lane : 0));
                const bool got = res_poll(a.abort_w, [&]() -> bool {
                    if (!h0) h0 = ld_gran(in, tag, t0);
                    if (!h1) h1 = ld_gran(in + 2, tag, t1);
                    if (!h2) h2 = ld_gran(in + 4, tag, t2);
                    return h0 && h1 && h2;
                });
                if (!got) *flag_s = 0;
                // (members in lane order, fixed tree)
                t0 = wave_sum_dpp(t0);
                t1 = wave_sum_dpp(t1);
                t2 = wave_sum_dpp(t2);
                if (lane == 0) {
                    u64 *out = a.xscalg + 8 * cg;
                    st_gran(out, tag, t0);
                    st_gran(out + 2, tag, t1);
                    st_gran(out + 4, tag, t2);
                }
            }
            double u0 = 0.0, u1 = 0.0, u2 = 0.0;
            bool h0 = !(lane < ncl), h1 = h0, h2 = h0;
            u64 *in = a.xscalg + 8 * (lane < ncl ? lane : 0);
            const bool got = res_poll(a.abort_w, [&]() -> bool {
                if (!h0) h0 = ld_gran(in, tag, u0);
                if (!h1) h1 = ld_gran(in + 2, tag, u1);
                if (!h2) h2 = ld_gran(in + 4, tag, u2);
                return h0 && h1 && h2;
            });
            if (!got) *flag_s = 0;
            u0 = wave_sum_dpp(u0);
            u1 = wave_sum_dpp(u1);
            u2 = wave_sum_dpp(u2);
            // (red[8..10]: the block sums use red[0..7], and slow waves may still be reading those)
            if (lane == 0) {
                red[8] = u0;
                red[9] = u1;
                red[10] = u2;
            }
        }
        tick(15);
        __syncthreads();
        Rtot = red[8];
        pp1 = red[9];
        pp0 = red[10];
        ++ng;
        tick(9);
        return *flag_s != 0;
    };

    auto finish = [&](int k_run) {
        __syncthreads();
        if (tid < nc)
            for (int c = 0; c < a.C; ++c) {
                a.x_cur[(int64_t)c * M + j0 + tid] = xc_all[c * cpw + tid];
                if (a.gcur_io) a.gcur_io[(int64_t)c * M + j0 + tid] = gc_all[c * cpw + tid];
            }
        if (w == 0 && a.ucur_io && tid < 3 * a.C) a.ucur_io[tid] = ucs[tid];
        if (w == 0 && tid == 0) {
            a.n_run[0] = k_run;
            a.n_run[1] = ev;
            a.n_run[2] = ng;
        }
        if (timing) {
            tick(10);
            for (int i = 0; i < 16; ++i) a.dbg[(w == 0 ? 0 : 16) + i] += tacc_s[i];
        }
    };

    // ---- potential and gradient at the current sample of every chain (unless kept from the
    // previous launch)
    if (!a.have_state) {
        for (int c = 0; c < a.C; ++c) {
            __syncthreads();
            if (tid < nc) xs[tid] = xc_all[c * cpw + tid];
            if (!evaluate(true)) return;
            dots(0, 0.0);
            __syncthreads();
            if (tid < nc) gc_all[c * cpw + tid] = gs[tid];
            if (!gather_scalars(Rw, 0.0, 0.0)) return;
            if (tid == 0) {
                ucs[3 * c + 0] = ud + a.alpha * Rtot;
                ucs[3 * c + 1] = ud;
                ucs[3 * c + 2] = Rtot;
            }
        }
    }
    __syncthreads();

    long long accepts = a.accept_count0;
    int k = 0;
    for (; k < a.K; ++k) {
        const int Lk = a.L[k];
        const double u = a.us[k];
        const int c = a.chain ? a.chain[k] : 0;
        double *xc = xc_all + c * cpw, *gc = gc_all + c * cpw;
        // momentum of this trajectory, first half step from the kept gradient (hmc.py:95-113)
        double q = 0.0;
        __syncthreads();
        if (tid < nc) {
            const double p0 = a.p0s[(int64_t)k * M + j0 + tid];
            q = p0 * p0;
            double pj = p0 - 0.5 * a.dt * gc[tid];
            double xj = xc[tid] + a.dt * pj;
            if (xj > hi[tid]) {
                xj = hi[tid];
                pj = -pj;
            } else if (xj < lo[tid]) {
                xj = lo[tid];
                pj = -pj;
            }
            ps[tid] = pj;
            xs[tid] = xj;
        }
        const double pp0w = res_block_sum(q, red);
        for (int s = 0; s < Lk; ++s) {
            if (s > 0) dots(1, a.dt);
            if (!evaluate(s == Lk - 1)) return;
        }
        const double pp1w = dots(2, 0.5 * a.dt);
        if (!gather_scalars(Rw, pp1w, pp0w)) return;
        double Ucur = ucs[3 * c + 0], Ucur_d = ucs[3 * c + 1], Ucur_r = ucs[3 * c + 2];
        const double Unew = ud + a.alpha * Rtot;
        const double Hcur = 0.5 * pp0 + Ucur;
        const double Hnew = 0.5 * pp1 + Unew;
        const bool acc = (Hnew < Hcur) || (u < exp(-(Hnew - Hcur)));  // hmc.py:158-177
        __syncthreads();  // every thread has read ucs
        if (acc) {
            Ucur = Unew;
            Ucur_d = ud;
            Ucur_r = Rtot;
            accepts += 1;
            if (tid == 0) {
                ucs[3 * c + 0] = Ucur;
                ucs[3 * c + 1] = Ucur_d;
                ucs[3 * c + 2] = Ucur_r;
            }
            if (tid < nc) {
                xc[tid] = xs[tid];
                gc[tid] = gs[tid];
                if (a.xacc) a.xacc[(int64_t)k * M + j0 + tid] = xs[tid];
            }
        }
        if (w == 0 && tid == 0) {
            a.accepted[k] = acc ? 1 : 0;
            double *o = a.out5s + 5 * k;
            o[0] = Ucur;
            o[1] = Ucur_d;
            o[2] = Ucur_r;
            o[3] = Hcur;
            o[4] = Hnew;
        }
        if (acc && a.stop_at_accepts > 0 && accepts >= a.stop_at_accepts) {
            ++k;
            break;
        }
    }
    finish(k);
}

}  // namespace ghk
