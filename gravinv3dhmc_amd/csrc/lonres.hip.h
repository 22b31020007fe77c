// The longitude-harmonic pass of the shift-invariant store as ONE persistent launch per batch of trajectories
// (gfx950): lonsymh.hip.h's leapfrog step is four dependent, latency-bound launches (R^ -> sweep -> sum of the
// sweep's partials -> epilogue: 48 us of kernels + their gaps at C4, the table read from the Infinity Cache in
// every step).  Here a workgroup keeps ITS cell rows' part of the table T^ in REGISTERS for the whole launch
// (C4: 35.7 MB over 200 workgroups -- the chip's register files hold it), the chain's state of its cells with
// it, and the workgroups exchange per evaluation what the launches exchanged through memory:
//
//   forward product of the workgroup's rows: D^ partial (na x nf complex)  -> slab row, untagged, one flag
//   hop 1   the workgroups of a cluster (same XCD) sum a chunk of the entries over the cluster's members
//           -> cluster sums, tagged granules ("the data is the flag", resident.hip.h)
//   hop 2   the OWNER of observation class a (na of the workgroups) sums the 8 clusters' D^[a][:], inverse
//           transform -> the class's predicted data; the classes' sums of d meet in a one-hop all-gather (the
//           mean of potential.py:700-706); residuals, |r|^2 share, forward transform -> R^[a][:] published
//   hop 3   every workgroup collects R^ (59.5 KB at C4) and the scalars: U_data, the regulariser's value,
//           the kinetic energy of the trajectory's momentum
//   adjoint product, gradient, leapfrog update (hmc.py:114-152) -- and round again.
//
// Trajectories end inside the launch: the final momentum's sum of squares in a one-hop all-gather, the
// Metropolis test (hmc.py:159-177) evaluated identically by every workgroup.  Same contract as
// resident_chain_kernel (K trajectories per launch, the host draws the momenta and variates in the reference's
// order; every wait bounded, an aborted launch leaves the chain untouched).  All four regularisers: the element-wise
// ones (Damping, MS) behind the publish flag; the stencil kinds (Smoothness, TV) read their neighbours' cells from
// the positions every workgroup publishes (write-through) with an evaluation, behind hop 3, and their value
// travels with the trajectory's p'p.
// Reference arithmetic: inversion/hmc.py:85-177, inversion/potential.py:698-736, gravmag/_tesseroid_numba.py:207-222.
#pragma once
#include "lonsymh.hip.h"
#include "resbatch.hip.h"

namespace ghk {

constexpr int LR_THREADS = 256;
constexpr int LR_MAXWG = 256;

struct LonResArgs {
    LonHarmGeom g;
    int64_t N, M;
    int nwg, try_local;
    const double *wm;             // column weights (nullptr: none)
    const double *low, *high, *mwapr, *wm2;
    int kind, ms_grad_den_mw;
    int nz, ny, nx;               // the model's shape (stencil regularisers)
    double alpha, beta;
    double *xpub;                 // 2 x M: the models as the stencil regularisers see them (Smoothness, TV), by parity
    const double *dobs_c, *gfix;  // gfix: nullptr without a fixed part of the data term
    double gfix_sum;
    const d2 *Mhat;               // [na][nf]: transform of the slots' observation counts (R^ of a residual of ones)
    // chain
    double *x_cur;                // M: in = current sample, out = current sample after the last trajectory run
    int K;
    const int *L;
    const double *p0s;            // K x M
    const double *us;             // K
    double dt;
    long long stop_at_accepts, accept_count0;
    int *accepted;                // K
    double *out5s;                // K x {U, U_data, R, H_current, H_proposal}
    double *xacc;                 // K x M accepted samples (nullptr: not wanted)
    int *n_run;                   // [0] trajectories run, [1] evaluations, [2] trajectory-end exchanges
    double *ucur;                 // 3: {U, U_data, R} of the current sample at the end
    // exchange (tags continue over launches)
    d2 *slab;                     // (nwg + 8) x E forward partials, untagged
    u64 *flagg;                   // nwg + 8
    u32x4 *xslabg;                // 2 x 8 x E x 2 cluster sums
    d2 *rhatg;                    // 2 x E, untagged: complete behind the classes' scalars
    u32x4 *clsg;                  // 2 x 64 x 4: per class {sum of d, sum of q, sum of q^2} (q: residual against the previous mean)
    u32x4 *scalg;                 // 2 x LR_MAXWG x 2: per workgroup {regulariser share, p0'p0 share}
    u32x4 *ppg;                   // 2 x LR_MAXWG x 2: per workgroup {p'p share, stencil regulariser's share} at the end of a trajectory
    u64 *xccg;
    unsigned tag0, tagE0, ltag;
    unsigned *abort_w;
    long long *dbg;               // optional: 16 accumulated phase times (100 MHz ticks) of workgroup 0
};

static inline size_t lonres_lds_doubles(int n, int nf, int na, int rw)
{
    return lonsymh_lds_doubles(n, nf, na, rw) + 2 * (size_t)rw + 2 * 64 + 128 + 8 * 64 + 2 * 64 + 3 * LR_MAXWG + 64;
}

// Sums of NV values over the workgroup's four waves in a fixed tree (the same bits wherever the same values meet),
// valid in every thread.  redn: 4 NV doubles of LDS; two barriers.
template <int NV>
__device__ __forceinline__ void block_sums(double (&v)[NV], double *redn)
{
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = wave_allreduce_sum(v[i]);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NV; ++i) redn[wave * NV + i] = v[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = ((redn[i] + redn[NV + i]) + redn[2 * NV + i]) + redn[3 * NV + i];
}

// X^[f] = sum_k x[k] e^{-2 pi i f k / n} of NR real sequences given as their even and odd parts over the pairs (k, n - k):
// ev[kp] = x[kp] + x[n - kp], od[kp] = x[kp] - x[n - kp] (kp = 0 and 2 kp = n: the cell itself, od = 0), kp in [k0, k1):
//     x[k] e^{-i t} + x[n - k] e^{+i t} = ev cos t - i od sin t  -- half the terms of the plain sum (lh_dft_part).
template <int NR>
__device__ __forceinline__ void lh_dft_eo_rows(const double *ev, const double *od, int stride, const d2 *tws, int k0, int k1, int f,
                                               int n, d2 (&acc)[NR])
{
#pragma unroll
    for (int r = 0; r < NR; ++r) acc[r] = d2{0.0, 0.0};
    int idx = (int)(((long long)f * k0) % n);
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
        d2 w[4];
        double e[NR][4], o[NR][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            w[u] = tws[idx];
            idx += f;
            if (idx >= n) idx -= n;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                e[r][u] = ev[r * stride + k + u];
                o[r][u] = od[r * stride + k + u];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                acc[r].x += e[r][u] * w[u].x;
                acc[r].y -= o[r][u] * w[u].y;
            }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; k < k1; ++k) {
        const d2 w = tws[idx];
        idx += f;
        if (idx >= n) idx -= n;
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            acc[r].x += ev[r * stride + k] * w.x;
            acc[r].y -= od[r * stride + k] * w.y;
        }
    }
}

// sum_f Re(H[f] e^{+2 pi i f k / n}) at k = kp and at k = n - kp from ONE pass: cs = sum_f H[f].x cos, sn = sum_f H[f].y sin
// (angle 2 pi f kp / n): the value at kp is cs - sn, at n - kp cs + sn.
__device__ __forceinline__ void lh_idft_pair(const d2 *H, const d2 *tws, int f0, int nf, int kp, int n, double &cs, double &sn)
{
    cs = 0.0;
    sn = 0.0;
    int idx = (int)(((long long)f0 * kp) % n), ff = f0;
    for (; ff + 8 <= nf; ff += 8) {
        d2 w[8], h[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            w[u] = tws[idx];
            idx += kp;
            if (idx >= n) idx -= n;
            h[u] = H[ff + u];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            cs += h[u].x * w[u].x;
            sn += h[u].y * w[u].y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; ff < nf; ++ff) {
        const d2 w = tws[idx], h = H[ff];
        idx += kp;
        if (idx >= n) idx -= n;
        cs += h.x * w.x;
        sn += h.y * w.y;
    }
}

template <int RW>
__global__ void __launch_bounds__(LR_THREADS) lonsymh_resident_kernel(LonResArgs a)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const LonHarmGeom &g = a.g;
    const int tid = threadIdx.x, f = tid & 63, ag = tid >> 6, lane = tid & 63, wave = tid >> 6;
    const int n = g.n, nf = g.nf, na = g.na, E = na * nf;
    const int w = blockIdx.x, nwg = a.nwg;
    const bool fv = f < nf;
    d2 *Rh = reinterpret_cast<d2 *>(smem);                   // na x nf
    d2 *tws = Rh + (size_t)na * nf;                           // n
    d2 *Gp = tws + n;                                         // RW x 4 x nf
    d2 *Gh = Gp + RW * 4 * nf;                                // RW x nf
    double *xs = reinterpret_cast<double *>(Gh + RW * nf);    // RW x nf even parts, RW x nf odd parts of the rows' x / w
    double *xo = xs + RW * nf;
    d2 *Xp = reinterpret_cast<d2 *>(xs + RW * (n + 2));       // RW x 4 x nf
    double *red = reinterpret_cast<double *>(Xp + RW * 4 * nf);  // 16
    d2 *Dh = reinterpret_cast<d2 *>(red + 16);                // 64: the owner's D^[a][:]
    double *row = reinterpret_cast<double *>(Dh + 64);        // 128: slot sums of the class's residuals
    d2 *Rp = reinterpret_cast<d2 *>(row + 128);               // 4 x 64 quarters of the owner's forward transform
    double *cls = reinterpret_cast<double *>(Rp + 4 * 64);    // 2 x 64 class scalars
    double *scs = cls + 2 * 64;                               // 3 x LR_MAXWG (block_sums' scratch in front)
    double *redn = scs;
    d2 *Cr = reinterpret_cast<d2 *>(scs + 64);                // RW x nf: sum_a conj(T^_r[a][f]) M^[a][f], see the mean
    int *flag_s = reinterpret_cast<int *>(scs + 3 * LR_MAXWG);
    long long *tacc_s = reinterpret_cast<long long *>(scs + 3 * LR_MAXWG + 2);

    long long tlast = 0;
    const bool timing = a.dbg != nullptr && tid == 0 && w == 0;
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

    // ---- resident operands: the rows' table (registers), twiddles (LDS), the cells' constants (registers)
    d2 th[RW][LH_AK];
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int c = w + r * nwg;
        const d2 *Tg = g.That + (int64_t)(c < g.nc ? c : w) * na * nf;
#pragma unroll
        for (int u = 0; u < LH_AK; ++u) {
            const int aa = ag * LH_AK + u;
            th[r][u] = (fv && aa < na && c < g.nc) ? Tg[aa * nf + f] : d2{0.0, 0.0};
        }
    }
    for (int e = tid; e < n; e += LR_THREADS) tws[e] = g.tw[e];
    if (tid == 0) *flag_s = 1;
    // items of a thread: the cells at the longitudes kp = lane and n - kp of row `wave` (a wave per row: RW <= 4) -- the
    // pair shares every term of both transforms (cos even, sin odd about the pair)
    constexpr int NI = 2;
    const int kp = lane, rt = wave;
    const bool pair_ok = rt < RW && kp < nf;
    int64_t ij[NI];
    bool iv[NI];
    double iw[NI], hi[NI], lo[NI], apr[NI], w2[NI], xc[NI], gc[NI], x[NI], p[NI], grad[NI];
#pragma unroll
    for (int q = 0; q < NI; ++q) {
        const int c = w + rt * nwg;
        const int kq = q == 0 ? kp : n - kp;
        iv[q] = pair_ok && c < g.nc && (q == 0 || (kp > 0 && 2 * kp != n));
        ij[q] = (int64_t)(iv[q] ? c : 0) * n + (iv[q] ? kq : 0);
        iw[q] = 1.0;
        hi[q] = lo[q] = apr[q] = xc[q] = gc[q] = x[q] = p[q] = grad[q] = 0.0;
        w2[q] = 1.0;
        if (iv[q]) {
            const int64_t j = ij[q];
            const double wj = a.wm ? a.wm[j] : 1.0;
            iw[q] = (wj != 0.0) ? 1.0 / wj : 1.0;
            hi[q] = a.high[j];
            lo[q] = a.low[j];
            apr[q] = a.mwapr[j];
            if (a.kind == 2) w2[q] = a.wm2[j];
            xc[q] = a.x_cur[j];
            x[q] = xc[q];
        }
    }

    // ---- clusters and placement (resident.hip.h)
    const int ncl = nwg < RES_CLUSTERS ? nwg : RES_CLUSTERS;
    const int cg = w % RES_CLUSTERS, crank = w / RES_CLUSTERS;
    const int cn = (nwg - cg + RES_CLUSTERS - 1) / RES_CLUSTERS;
    const int ch = (E + cn - 1) / cn;  // entries of the cluster sum this workgroup produces
    bool local = false;
    {
        const unsigned xcc = (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID[3:0]
        if (tid == 0)
            __hip_atomic_store(a.xccg + w, ((u64)a.ltag << 32) | xcc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
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
            if (lane == 0) red[8] = (got && __all(same) && a.try_local) ? 1.0 : 0.0;
            if (!got) *flag_s = 0;
        }
        __syncthreads();
        local = red[8] != 0.0;
        if (*flag_s == 0) return;
    }
    const __amdgpu_buffer_rsrc_t rs_slab = __builtin_amdgcn_make_buffer_rsrc(a.slab, 0, (int)((size_t)(nwg + 8) * E * 16), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xs = __builtin_amdgcn_make_buffer_rsrc(a.xslabg, 0, 2 * RES_CLUSTERS * E * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_rh = __builtin_amdgcn_make_buffer_rsrc(a.rhatg, 0, 2 * E * 16, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_cls = __builtin_amdgcn_make_buffer_rsrc(a.clsg, 0, 2 * 64 * 64, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_sc = __builtin_amdgcn_make_buffer_rsrc(a.scalg, 0, 2 * LR_MAXWG * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_pp = __builtin_amdgcn_make_buffer_rsrc(a.ppg, 0, 2 * LR_MAXWG * 32, 0x00020000);

    // ---- the class this workgroup owns (at most one: nwg >= na), the observations of its slots
    int own = -1;
    {
        const int a0 = (int)(((long long)w * na + nwg - 1) / nwg);
        if (a0 < na && (int)(((long long)a0 * nwg) / na) == w) own = a0;
    }
    int o_i[3] = {-1, -1, -1};
    double o_dobs[3] = {0.0, 0.0, 0.0}, o_gfix[3] = {0.0, 0.0, 0.0};
    if (own >= 0 && tid < n) {
        const int e = own * n + tid;
        o_i[0] = g.slot_first[e];
        int nx = 1;
        for (int xx = 0; xx < g.n_xslots; ++xx)
            if (g.xslot[xx] == e)
                for (int qq = g.xptr[xx]; qq < g.xptr[xx + 1] && nx < 3; ++qq) o_i[nx++] = g.xobs[qq];
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (o_i[u] >= 0) {
                o_dobs[u] = a.dobs_c[o_i[u]];
                o_gfix[u] = a.gfix ? a.gfix[o_i[u]] : 0.0;
            }
    }
    const int qn = (n + 3) / 4, qf = (nf + 3) / 4;
    const bool stencil = a.kind == 1 || a.kind == 3;
    RegArgs ra{};
    ra.ms_grad_den_mw = a.ms_grad_den_mw;
    ra.kind = a.kind;
    ra.M = a.M;
    ra.nz = a.nz;
    ra.ny = a.ny;
    ra.nx = a.nx;
    ra.alpha = a.alpha;
    ra.beta = a.beta;
    ra.mwapr = a.mwapr;
    ra.wm2 = a.wm2;
    __syncthreads();
    // What a change dm of the data's mean does to the adjoint product: r -> r - dm in every observation is
    // R^ -> R^ - dm M^, S^_r[f] -> S^_r[f] - dm C_r[f] with C_r[f] = sum_a conj(T^_r[a][f]) M^[a][f]: a constant of the row.
    if (fv) {
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            d2 gp = d2{0.0, 0.0};
#pragma unroll
            for (int u = 0; u < LH_AK; ++u) {
                const int aa = ag * LH_AK + u;
                const d2 m = a.Mhat[(aa < na ? aa : na - 1) * nf + f];
                gp.x += th[r][u].x * m.x + th[r][u].y * m.y;
                gp.y += th[r][u].x * m.y - th[r][u].y * m.x;
            }
            Gp[(r * 4 + ag) * nf + f] = gp;
        }
    }
    __syncthreads();
    for (int e = tid; e < RW * nf; e += LR_THREADS) {
        const int r = e / nf, ff = e - r * nf;
        d2 sres = Gp[(r * 4) * nf + ff];
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            sres.x += Gp[(r * 4 + q) * nf + ff].x;
            sres.y += Gp[(r * 4 + q) * nf + ff].y;
        }
        Cr[e] = sres;
    }
    __syncthreads();

    // ---- the chain
    // (k = -2: a first evaluation at the current sample for the mean of its data alone -- the residuals of an
    // evaluation are formed against the mean of the one before)
    int k = -2, s = 0, Lk = 0, n_done = 0;
    double mean_prev = 0.0, sc5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
    double uk = 0.0, pp0 = 0.0, pp0_share = 0.0, U0 = 0.0, U1 = 0.0, U2 = 0.0;
    long long accepts = a.accept_count0;
    unsigned ev = 0, evE = 0;
    bool stop = false;
    // one-hop all-gather of two scalars per workgroup (every workgroup reads every workgroup's pair; fixed-tree sums)
    auto gather2 = [&](double v0, double v1, double &s0, double &s1) -> bool {
        evE += 1;
        const unsigned tagE = a.tagE0 + evE;
        const int parE = (int)(evE & 1u);
        if (tid == 0) {
            rb_store(rs_pp, (unsigned)(((parE * LR_MAXWG + w) * 2 + 0) * 16), rb_pack(tagE, v0), false);
            rb_store(rs_pp, (unsigned)(((parE * LR_MAXWG + w) * 2 + 1) * 16), rb_pack(tagE, v1), false);
        }
        double gsum[2] = {0.0, 0.0};
        {
            unsigned off[2] = {(unsigned)(((parE * LR_MAXWG + (tid < nwg ? tid : 0)) * 2) * 16),
                               (unsigned)(((parE * LR_MAXWG + (tid < nwg ? tid : 0)) * 2 + 1) * 16)};
            double v[2];
            const bool got = rb_poll<2>(a.abort_w, rs_pp, tagE, tid < nwg ? 2 : 0, off, v);
            if (tid < nwg) {
                gsum[0] = v[0];
                gsum[1] = v[1];
            }
            if (!got) *flag_s = 0;
        }
        block_sums<2>(gsum, redn);
        s0 = gsum[0];
        s1 = gsum[1];
        return *flag_s != 0;
    };
    tick(0);
    for (;;) {
        ev += 1;
        const unsigned tag = a.tag0 + ev;
        const int par = (int)(ev & 1u);
        // ================= evaluation at x: forward product of the rows, the exchange, adjoint product -> grad
        if (stencil) {
            // (the neighbours' cells belong to other workgroups: write-through, read behind hop 3 -- every workgroup's
            // stores are drained before its flag, and no R^ exists before every flag does)
#pragma unroll
            for (int q = 0; q < NI; ++q)
                if (iv[q]) st_wt(a.xpub + (int64_t)par * a.M + ij[q], x[q]);
        }
        if (pair_ok) {
            const double x1 = iv[0] ? x[0] * iw[0] : 0.0, x2 = iv[1] ? x[1] * iw[1] : 0.0;
            xs[rt * nf + kp] = x1 + x2;
            xo[rt * nf + kp] = iv[1] ? x1 - x2 : 0.0;
        }
        __syncthreads();
        // X^_r[f]: quarter ag of the longitude pairs, then the four quarters; D^ partial of the workgroup
        if (fv) {
            const int k0 = ag * qf, k1 = (k0 + qf < nf) ? k0 + qf : nf;
            d2 xq[RW];
            lh_dft_eo_rows<RW>(xs, xo, nf, tws, k0 < nf ? k0 : nf, k1, f, n, xq);
#pragma unroll
            for (int r = 0; r < RW; ++r) Xp[(r * 4 + ag) * nf + f] = xq[r];
        }
        __syncthreads();
        d2 dacc[LH_AK];
#pragma unroll
        for (int u = 0; u < LH_AK; ++u) dacc[u] = d2{0.0, 0.0};
        if (fv) {
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                d2 xh = Xp[(r * 4) * nf + f];
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    xh.x += Xp[(r * 4 + q) * nf + f].x;
                    xh.y += Xp[(r * 4 + q) * nf + f].y;
                }
#pragma unroll
                for (int u = 0; u < LH_AK; ++u) {
                    dacc[u].x += th[r][u].x * xh.x - th[r][u].y * xh.y;
                    dacc[u].y += th[r][u].x * xh.y + th[r][u].y * xh.x;
                }
            }
        }
        tick(1);
        // ---- publish: the partial behind one flag, the scalars as granules
        if (fv) {
#pragma unroll
            for (int u = 0; u < LH_AK; ++u) {
                const int aa = ag * LH_AK + u;
                if (aa < na) rb_store2(rs_slab, (unsigned)(((size_t)w * E + aa * nf + f) * 16), dacc[u].x, dacc[u].y, local);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            if (local)
                __hip_atomic_store(a.flagg + w, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else
                __hip_atomic_store(a.flagg + w, (u64)tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // (behind the flag, while the partials travel) the regulariser (potential.py:719-736): alpha dR/dx per cell, this
        // workgroup's share of its value; with the trajectory's p0'p0 share as granules -- read in hop 3
        double greg[NI], rval = 0.0;
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const double v = x[q] - apr[q];
            double gq = 0.0, val = 0.0;
            if (stencil) {
                // (behind hop 3, below)
            } else if (a.kind == 0) {
                val = v * v;
                gq = 2.0 * v;
            } else {
                const double v2 = v * v, den = v2 + a.beta;
                val = (w2[q] * v2) / den;
                const double deng = a.ms_grad_den_mw ? x[q] * x[q] + a.beta : den;
                gq = (2.0 * a.beta * w2[q] * v) / (deng * deng);
            }
            greg[q] = iv[q] ? a.alpha * gq : 0.0;
            rval += iv[q] ? val : 0.0;
        }
        {
            const double rsh = block_allreduce_sum(rval, red, LR_THREADS / 64);
            if (tid == 0) {
                rb_store(rs_sc, (unsigned)(((par * LR_MAXWG + w) * 2 + 0) * 16), rb_pack(tag, rsh), false);
                rb_store(rs_sc, (unsigned)(((par * LR_MAXWG + w) * 2 + 1) * 16), rb_pack(tag, pp0_share), false);
            }
            pp0_share = 0.0;  // (p0'p0 of a trajectory rides on its first evaluation only)
        }
        tick(2);
        // ---- hop 1: this workgroup's chunk of the entries, summed over the cluster's members in order
        if (wave == 0) {
            const bool got = res_poll(a.abort_w, [&]() -> bool {
                bool ok = true;
                if (lane < cn) {
                    const u64 e = __hip_atomic_load(a.flagg + cg + RES_CLUSTERS * lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = (unsigned)e == tag;
                }
                return ok;
            });
            if (!got) *flag_s = 0;
        }
        __syncthreads();
        if (*flag_s == 0) return;
        tick(3);
        for (int t0 = 0; t0 < ch; t0 += LR_THREADS) {
            const int e = crank * ch + t0 + tid;
            if (t0 + tid < ch && e < E) {
                d2 sum = d2{0.0, 0.0};
                int m = 0;
                for (; m + 8 <= cn; m += 8) {
                    d2 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        v[u] = rb_load2(rs_slab, (unsigned)(((size_t)(cg + RES_CLUSTERS * (m + u)) * E + e) * 16));
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        sum.x += v[u].x;
                        sum.y += v[u].y;
                    }
                }
                for (; m < cn; ++m) {
                    const d2 v = rb_load2(rs_slab, (unsigned)(((size_t)(cg + RES_CLUSTERS * m) * E + e) * 16));
                    sum.x += v.x;
                    sum.y += v.y;
                }
                const unsigned off = (unsigned)((((size_t)par * RES_CLUSTERS + cg) * E + e) * 32);
                rb_store(rs_xs, off, rb_pack(tag, sum.x), false);
                rb_store(rs_xs, off + 16, rb_pack(tag, sum.y), false);
            }
        }
        tick(4);
        // ---- hop 2: the owner of a class
        if (own >= 0) {
            if (wave == 0) {
                unsigned off[16];
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    off[u] = (unsigned)((((size_t)par * RES_CLUSTERS + (u >> 1)) * E + own * nf + (fv ? f : 0)) * 32 + (u & 1) * 16);
                const bool got = rb_poll<16>(a.abort_w, rs_xs, tag, fv ? 2 * ncl : 0, off, v);
                if (!got) *flag_s = 0;
                if (fv) {
                    double sx = 0.0, sy = 0.0;
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (u < ncl) {
                            sx += v[2 * u];
                            sy += v[2 * u + 1];
                        }
                    const double wf = (f == 0 || (2 * f == n)) ? 1.0 : 2.0;
                    Dh[f] = d2{sx * wf, sy * wf};
                }
            }
            __syncthreads();
            if (*flag_s == 0) return;
            tick(10);
            // the class's predicted data; residuals against the PREVIOUS evaluation's mean (potential.py:700-706:
            // r = d + grav_fix - mean - dobs; the mean of this evaluation needs every class -- the consumers correct for
            // the difference, below), slot sums, the class's sums
            double ds = 0.0, q1 = 0.0, q2 = 0.0;
            {
                // (the pair (k, n - k) from one pass, the frequencies in four quarters: thread (kp = lane, quarter ag))
                double *csq = reinterpret_cast<double *>(Rp), *snq = csq + 4 * 64;
                if (fv) {
                    const int f0 = ag * qf, f1 = (f0 + qf < nf) ? f0 + qf : nf;
                    double c_, s_;
                    lh_idft_pair(Dh, tws, f0 < nf ? f0 : nf, f1, f, n, c_, s_);
                    csq[ag * 64 + f] = c_;
                    snq[ag * 64 + f] = s_;
                }
            }
            __syncthreads();
            if (tid < n) {
                const double *csq = reinterpret_cast<const double *>(Rp), *snq = csq + 4 * 64;
                const int kq = (2 * tid <= n) ? tid : n - tid;
                const double c_ = ((csq[kq] + csq[64 + kq]) + csq[128 + kq]) + csq[192 + kq];
                const double s_ = ((snq[kq] + snq[64 + kq]) + snq[128 + kq]) + snq[192 + kq];
                const double dk = ((2 * tid <= n) ? c_ - s_ : c_ + s_) / (double)n;
                double rs_ = 0.0;
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (o_i[u] >= 0) {
                        const double qi = ((dk + o_gfix[u]) - mean_prev) - o_dobs[u];
                        ds += dk;
                        rs_ += qi;
                        q1 += qi;
                        q2 += qi * qi;
                    }
                row[tid] = rs_;
            }
            double cs[3] = {ds, q1, q2};
            block_sums<3>(cs, redn);  // (barriers: row is in place)
            tick(11);
            if (fv) {
                // R^[a][f] from the slot sums' even and odd parts over the pairs (k, n - k): quarter ag of the pairs
                const int k0 = ag * qf, k1 = (k0 + qf < nf) ? k0 + qf : nf;
                d2 rq = d2{0.0, 0.0};
                int idx = (int)(((long long)f * k0) % n);
                for (int kk = k0; kk < k1; ++kk) {
                    const bool self = (kk == 0) || (2 * kk == n);
                    const double v1 = row[kk], v2 = self ? 0.0 : row[n - kk];
                    const d2 wv = tws[idx];
                    idx += f;
                    if (idx >= n) idx -= n;
                    rq.x += (v1 + v2) * wv.x;
                    rq.y -= (self ? 0.0 : v1 - v2) * wv.y;
                }
                Rp[ag * 64 + f] = rq;
            }
            __syncthreads();
            if (tid < nf) {
                d2 sres = Rp[tid];
#pragma unroll
                for (int u = 1; u < 4; ++u) {
                    sres.x += Rp[u * 64 + tid].x;
                    sres.y += Rp[u * 64 + tid].y;
                }
                rb_store2(rs_rh, (unsigned)(((size_t)par * E + own * nf + tid) * 16), sres.x, sres.y, false);
            }
            // (R^[a][:] is untagged: complete once the class's scalars -- written behind the storing wave's drain and a
            // barrier -- carry the tag)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid < 3) rb_store(rs_cls, (unsigned)(((par * 64 + own) * 4 + tid) * 16), rb_pack(tag, cs[tid]), false);
        }
        tick(5);
        // ---- hop 3: R^ and the scalars, everybody
        {
            bool okall = true;
            double v5[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
            {
                unsigned off[3];
                double v[3];
#pragma unroll
                for (int u = 0; u < 3; ++u) off[u] = (unsigned)(((par * 64 + (tid < na ? tid : 0)) * 4 + u) * 16);
                okall = rb_poll<3>(a.abort_w, rs_cls, tag, tid < na ? 3 : 0, off, v) && okall;
                if (tid < na) {
                    v5[2] = v[0];
                    v5[3] = v[1];
                    v5[4] = v[2];
                }
            }
            {
                unsigned off[2] = {(unsigned)(((par * LR_MAXWG + (tid < nwg ? tid : 0)) * 2) * 16),
                                   (unsigned)(((par * LR_MAXWG + (tid < nwg ? tid : 0)) * 2 + 1) * 16)};
                double v[2];
                okall = rb_poll<2>(a.abort_w, rs_sc, tag, tid < nwg ? 2 : 0, off, v) && okall;
                if (tid < nwg) {
                    v5[0] = v[0];
                    v5[1] = v[1];
                }
            }
            if (!okall) *flag_s = 0;
            // (sums over the workgroups / classes in a fixed tree: the same bits in every workgroup; the barriers inside
            // also publish R^ and the flag)
            block_sums<5>(v5, redn);
            if (*flag_s != 0) {
                // every class's scalars carry the tag: R^ is complete (59.5 KB at C4, write-through stores, L2-served loads)
                for (int e0 = 0; e0 < E; e0 += 8 * LR_THREADS) {
                    d2 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u * LR_THREADS + tid;
                        v[u] = rb_load2(rs_rh, (unsigned)(((size_t)par * E + (e < E ? e : 0)) * 16));
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u * LR_THREADS + tid;
                        if (e < E) Rh[e] = v[u];
                    }
                }
            }
            __syncthreads();
            sc5[0] = v5[0];
            sc5[1] = v5[1];
            sc5[2] = v5[2];
            sc5[3] = v5[3];
            sc5[4] = v5[4];
        }
        if (*flag_s == 0) return;
        tick(6);
        double Rtot = sc5[0];
        const double pp0tot = sc5[1];
        const double mean = (sc5[2] + a.gfix_sum) / (double)a.N;
        const double dmean = mean - mean_prev;
        const double Ud = (sc5[4] - 2.0 * dmean * sc5[3]) + (double)a.N * dmean * dmean;
        mean_prev = mean;
        double rshare_st = 0.0;
        if (stencil) {
            // Smoothness / TV (potential.py:786-810) at the positions every workgroup published with this evaluation
            ra.x = a.xpub + (int64_t)par * a.M;
            double rv = 0.0;
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                double val = 0.0;
                greg[q] = iv[q] ? a.alpha * reg_cell<false, true>(ra, ij[q], x[q], val) : 0.0;
                rv += iv[q] ? val : 0.0;
            }
            rshare_st = block_allreduce_sum(rv, red, LR_THREADS / 64);  // (its total travels with the trajectory's p'p)
        }
        // ---- adjoint product: S^_r[f] = sum_a conj(T^_r[a][f]) R^[a][f]
        if (fv) {
            d2 rr[LH_AK];
#pragma unroll
            for (int u = 0; u < LH_AK; ++u) {
                const int aa = ag * LH_AK + u;
                rr[u] = Rh[(aa < na ? aa : na - 1) * nf + f];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                d2 gp = d2{0.0, 0.0};
#pragma unroll
                for (int u = 0; u < LH_AK; ++u) {
                    gp.x += th[r][u].x * rr[u].x + th[r][u].y * rr[u].y;
                    gp.y += th[r][u].x * rr[u].y - th[r][u].y * rr[u].x;
                }
                Gp[(r * 4 + ag) * nf + f] = gp;
            }
        }
        tick(13);
        __syncthreads();
        for (int e = tid; e < RW * nf; e += LR_THREADS) {
            const int r = e / nf, ff = e - r * nf;
            d2 sres = Gp[(r * 4) * nf + ff];
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                sres.x += Gp[(r * 4 + q) * nf + ff].x;
                sres.y += Gp[(r * 4 + q) * nf + ff].y;
            }
            const double wf = (ff == 0 || (2 * ff == n)) ? 1.0 : 2.0;
            Gh[e] = d2{(sres.x - dmean * Cr[e].x) * wf, (sres.y - dmean * Cr[e].y) * wf};
        }
        __syncthreads();
        tick(14);
        {
            double cs, sn;
            lh_idft_pair(Gh + (rt < RW ? rt : 0) * nf, tws, 0, nf, kp < nf ? kp : 0, n, cs, sn);
            if (iv[0]) grad[0] = 2.0 * (((cs - sn) / (double)n) * iw[0]) + greg[0];
            if (iv[1]) grad[1] = 2.0 * (((cs + sn) / (double)n) * iw[1]) + greg[1];
        }
        tick(7);
        // ================= what the evaluation was for
        bool next_traj = false;
        if (k == -2) {
            k = -1;  // (the mean is known now: the same evaluation again, for good)
        } else if (k < 0) {
            // the chain's current sample: gradient and potential (hmc.py:85-93)
#pragma unroll
            for (int q = 0; q < NI; ++q) gc[q] = grad[q];
            if (stencil) {
                double unused = 0.0;
                if (!gather2(0.0, rshare_st, unused, Rtot)) return;
            }
            U1 = Ud;
            U2 = Rtot;
            U0 = Ud + a.alpha * Rtot;
            next_traj = true;
        } else if (s < Lk) {
            // a full step (hmc.py:114-141)
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                double pj = p[q] - a.dt * grad[q];
                double xj = x[q] + a.dt * pj;
                if (xj > hi[q]) {
                    xj = hi[q];
                    pj = -pj;
                } else if (xj < lo[q]) {
                    xj = lo[q];
                    pj = -pj;
                }
                p[q] = pj;
                x[q] = xj;
            }
            if (s == 1) pp0 = pp0tot;  // (the trajectory's p0'p0 arrived with its first evaluation)
            s += 1;
        } else {
            if (s == 1) pp0 = pp0tot;
            // the last half step of the momentum, its sum of squares over all cells (hmc.py:151-157)
            double pps = 0.0;
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const double pf = p[q] - 0.5 * a.dt * grad[q];
                pps += iv[q] ? pf * pf : 0.0;
            }
            const double ppw = block_allreduce_sum(pps, red, LR_THREADS / 64);
            double pp1 = 0.0, rst = 0.0;
            if (!gather2(ppw, rshare_st, pp1, rst)) return;
            if (stencil) Rtot = rst;
            // Metropolis (hmc.py:159-177): the same bits in every workgroup
            const double Unew = Ud + a.alpha * Rtot;
            const double Hcur = 0.5 * pp0 + U0, Hnew = 0.5 * pp1 + Unew;
            const bool acc = (Hnew < Hcur) || (uk < exp(-(Hnew - Hcur)));
            if (acc) {
#pragma unroll
                for (int q = 0; q < NI; ++q) {
                    xc[q] = x[q];
                    gc[q] = grad[q];
                    if (a.xacc && iv[q]) a.xacc[(int64_t)k * a.M + ij[q]] = x[q];
                }
                U0 = Unew;
                U1 = Ud;
                U2 = Rtot;
                accepts += 1;
            }
            if (w == 0 && tid == 0) {
                a.accepted[k] = acc ? 1 : 0;
                double *o = a.out5s + 5 * (int64_t)k;
                o[0] = U0;
                o[1] = U1;
                o[2] = U2;
                o[3] = Hcur;
                o[4] = Hnew;
            }
            n_done = k + 1;
            if (acc && a.stop_at_accepts > 0 && accepts >= a.stop_at_accepts) stop = true;
            next_traj = true;
        }
        tick(8);
        if (next_traj) {
            k += 1;
            if (k >= a.K || stop) break;
            // the next trajectory: its momentum, first half step from the current sample (hmc.py:95-113)
            Lk = a.L[k];
            uk = a.us[k];
            double pps = 0.0;
#pragma unroll
            for (int q = 0; q < NI; ++q) {
                const double p0 = iv[q] ? a.p0s[(int64_t)k * a.M + ij[q]] : 0.0;
                pps += p0 * p0;
                double pj = p0 - 0.5 * a.dt * gc[q];
                double xj = xc[q] + a.dt * pj;
                if (xj > hi[q]) {
                    xj = hi[q];
                    pj = -pj;
                } else if (xj < lo[q]) {
                    xj = lo[q];
                    pj = -pj;
                }
                p[q] = pj;
                x[q] = xj;
            }
            pp0_share = block_allreduce_sum(pps, red, LR_THREADS / 64);  // this workgroup's share: published with the evaluation
            s = 1;
            tick(9);
        }
    }
    // ---- the chain's state
#pragma unroll
    for (int q = 0; q < NI; ++q)
        if (iv[q]) a.x_cur[ij[q]] = xc[q];
    if (w == 0 && tid == 0) {
        a.n_run[0] = n_done;
        a.n_run[1] = (int)ev;
        a.n_run[2] = (int)evE;
        a.ucur[0] = U0;
        a.ucur[1] = U1;
        a.ucur[2] = U2;
    }
    if (timing)
        for (int i = 0; i < 16; ++i) a.dbg[i] += tacc_s[i];
}

}  // namespace ghk
