// The shift-invariant store of a regular spherical grid in the LONGITUDE-HARMONIC domain (gfx950).
//
// lonsym.hip.h keeps K[i, (c, k)] = T[c][a_i][(m_i - k) mod n] and evaluates forward and adjoint as circular
// correlations along the longitude, directly: N M multiply-adds each (C4: 5.3e8).  A circular correlation of
// length n is a product per frequency of the length-n DFTs (real input: nf = n / 2 + 1 frequencies):
//     S^[c][f]  = sum_a conj(T^[c][a][f]) R^[a][f]          adjoint  (R = slot sums of the residual)
//     D^[a][f] += T^[c][a][f] X^[c][f]                       forward  (X = xs of the cell row)
// -- n_c n_a nf complex multiply-adds per product (C4: 2.2e6), ~60x less arithmetic, from a table of the same
// size (T^[c][a][f], complex: n_c n_a nf 16 bytes = 35.7 MB at C4), read once per leapfrog step.  The
// clamp-and-reflect update of hmc.py:135-141 is per cell, so a cell row's gradient goes back to the
// longitudes (inverse DFT of nf coefficients at n points), is updated, and the new positions are
// transformed again (DFT) for the forward product: all inside the workgroup that owns the cell row, which
// holds R^ (59.5 KB at C4) in LDS and its row of T^ in registers -- thread (f, group of classes).
// The lengths are small (n = 120): the transforms are direct sums against a table of the n twiddle factors
// (exact index arithmetic (f k) mod n; an FFT would save nothing at 4 barriers per cell row).
//
// Around the sweep: lonsymh_rhat_kernel (slot sums of r and their DFT, one block per class) in front of it,
// lonsymh_post_kernel behind it (sum of the workgroups' D^ partials, inverse DFT per class, scatter to the
// observations of the class: ONE finished slab row + per-class sums for the one-launch epilogue).
//
// Same values as lonsym.hip.h up to the rounding of the transforms (~1e-15 of a row's largest entry);
// sums in a fixed order: reproducible bit for bit.  Reference arithmetic: gravmag/_tesseroid_numba.py:207-222
// (cos(lon - lon')), gravmag/tesseroid.py:189-232, inversion/potential.py:698,708, inversion/hmc.py:114-152.
#pragma once
#include "lonsym.hip.h"
#include "resident.hip.h"

namespace ghk {

struct LonHarmGeom {
    int n, nf, na, nc;
    const d2 *That;          // [nc][na][nf]
    const d2 *tw;            // n: (cos, sin)(2 pi j / n)
    d2 *Rhat;                // [na][nf]
    d2 *Dpart;               // [grid][na][nf]
    const int *slot_first;   // na * n: first observation of slot (a, m), or -1
    const int *slot_x;       // na * n: the slot's entry in xptr (slots holding several observations), or -1
    int n_xslots;
    const int *xslot, *xptr, *xobs;
    int64_t N;
    long long *dbg;          // optional: 8 accumulated phase times (100 MHz ticks) of workgroup 0, thread 0
};

constexpr int LH_THREADS = 256;
constexpr int LH_AK = 16;  // classes per thread: na <= 4 * LH_AK; nf <= 64

static inline size_t lonsymh_lds_doubles(int n, int nf, int na, int rw)
{
    return 2 * (size_t)na * nf + 2 * (size_t)n + (size_t)rw * (8 * (size_t)nf + 2 * (size_t)nf + (size_t)n + 8 * (size_t)nf) + 16;
}

// T^[c][a][f] = sum_delta T[c][a][delta] e^{-2 pi i f delta / n}: one block per (c, a) row of the table
// (nfp: pitch of a row of T^ in complex entries, >= nf; the entries past nf are written as zeros)
// (rowmap: row c of T^ comes from row rowmap[c] of T, or nullptr: from row c)
__global__ void __launch_bounds__(64) lonsymh_table_kernel(const double *T, int64_t ldT, int n, int nf, int na, const d2 *tw, d2 *That,
                                                           int nfp, const int *rowmap = nullptr)
{
    __shared__ double row[1024];
    const int c = blockIdx.x / na, a = blockIdx.x - c * na;
    const double *src = T + (int64_t)(rowmap ? rowmap[c] : c) * ldT + (int64_t)a * n;
    for (int e = threadIdx.x; e < n; e += 64) row[e] = src[e];
    __syncthreads();
    for (int f = threadIdx.x; f < nf; f += 64) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int d = 0; d < n; ++d) {
            const d2 w = tw[idx];
            re += row[d] * w.x;
            im -= row[d] * w.y;
            idx += f;
            if (idx >= n) idx -= n;
        }
        That[((int64_t)c * na + a) * nfp + f] = d2{re, im};
    }
    for (int f = nf + threadIdx.x; f < nfp; f += 64) That[((int64_t)c * na + a) * nfp + f] = d2{0.0, 0.0};
}

__global__ void __launch_bounds__(256) lonsymh_twiddle_kernel(int n, d2 *tw)
{
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j < n) {
        double s, c;
        sincospi(2.0 * (double)j / (double)n, &s, &c);
        tw[j] = d2{c, s};
    }
}

// sum over ff in [f0, f1) of Re(H[ff] e^{+2 pi i ff k / n}): H and the twiddle table in LDS, reads in batches of
// eight in front of their arithmetic (one wave per SIMD: a read-use-read-use loop runs at LDS latency)
__device__ __forceinline__ double lh_idft_part(const d2 *H, const d2 *tws, int f0, int f1, int k, int n)
{
    double s = 0.0;
    int idx = (int)(((long long)f0 * k) % n);
    int ff = f0;
    for (; ff + 8 <= f1; ff += 8) {
        d2 h[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h[u] = H[ff + u];
            w[u] = tws[idx];
            idx += k;
            if (idx >= n) idx -= n;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) s += h[u].x * w[u].x - h[u].y * w[u].y;
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; ff < f1; ++ff) {
        const d2 h = H[ff], w = tws[idx];
        s += h.x * w.x - h.y * w.y;
        idx += k;
        if (idx >= n) idx -= n;
    }
    return s;
}

// sum over k in [k0, k1) of v[k] e^{-2 pi i f k / n}
__device__ __forceinline__ d2 lh_dft_part(const double *v, const d2 *tws, int k0, int k1, int f, int n)
{
    d2 acc = d2{0.0, 0.0};
    int idx = (int)(((long long)f * k0) % n);
    int k = k0;
    for (; k + 8 <= k1; k += 8) {
        double x[8];
        d2 w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = v[k + u];
            w[u] = tws[idx];
            idx += f;
            if (idx >= n) idx -= n;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc.x += x[u] * w[u].x;
            acc.y -= x[u] * w[u].y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; k < k1; ++k) {
        const d2 w = tws[idx];
        acc.x += v[k] * w.x;
        acc.y -= v[k] * w.y;
        idx += f;
        if (idx >= n) idx -= n;
    }
    return acc;
}

// R^[a][f] of the residual r: block = class a
__global__ void __launch_bounds__(256) lonsymh_rhat_kernel(LonHarmGeom g, const double *__restrict__ r)
{
    __shared__ double row[1024];
    __shared__ d2 tws[1024];
    __shared__ d2 part[4][64];
    const int a = blockIdx.x, n = g.n, nf = g.nf, tid = threadIdx.x;
    for (int m = tid; m < n; m += 256) {
        const int e = a * n + m, idx = g.slot_first[e];
        double v = idx >= 0 ? r[idx] : 0.0;
        const int x = g.slot_x[e];
        if (x >= 0)
            for (int q = g.xptr[x]; q < g.xptr[x + 1]; ++q) v += r[g.xobs[q]];
        row[m] = v;
        tws[m] = g.tw[m];
    }
    __syncthreads();
    const int f = tid & 63, q = tid >> 6, qn = (n + 3) / 4;
    if (f < nf) {
        const int k0 = q * qn, k1 = (k0 + qn < n) ? k0 + qn : n;
        part[q][f] = lh_dft_part(row, tws, k0 < n ? k0 : n, k1, f, n);
    }
    __syncthreads();
    if (tid < nf) {
        d2 sres = part[0][tid];
#pragma unroll
        for (int u = 1; u < 4; ++u) {
            sres.x += part[u][tid].x;
            sres.y += part[u][tid].y;
        }
        g.Rhat[a * nf + tid] = sres;
    }
}

// The fused pass (modes of SweepArgs as lonsym_sweep_kernel).  Thread (f = tid & 63, ag = tid >> 6) holds the
// classes ag * LH_AK + u of its frequency.  A workgroup works on RW cell rows AT ONCE (rows blockIdx.x +
// r * gridDim.x): all their T^ rows are requested up front (RW x 16 loads of 16 bytes per thread in flight:
// the table arrives in one burst, at the memory system's rate) and every phase -- products, transforms,
// update -- covers the RW rows between two barriers: four barriers per group of rows instead of five per
// row, and no phase waits for a row's data.  (One row at a time, the next row's table prefetched: 25 us per
// pass at C4; the phases of a single row are too short to hide an LDS or memory round trip each.)
template <int RW>
__global__ void __launch_bounds__(LH_THREADS) lonsymh_sweep_kernel(LonHarmGeom g, SweepArgs a, const double *__restrict__ wm)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, f = tid & 63, ag = tid >> 6;
    const int n = g.n, nf = g.nf, na = g.na;
    const int mode = a.mode;
    d2 *Rh = reinterpret_cast<d2 *>(smem);                   // na x nf
    d2 *tws = Rh + (size_t)na * nf;                           // n
    d2 *Gp = tws + n;                                         // RW x 4 x nf partial S^ of the class groups
    d2 *Gh = Gp + RW * 4 * nf;                                // RW x nf
    double *xs = reinterpret_cast<double *>(Gh + RW * nf);    // RW x n
    d2 *Xp = reinterpret_cast<d2 *>(xs + RW * n);             // RW x 4 x nf partial X^ of the longitude quarters
    double *red = reinterpret_cast<double *>(Xp + RW * 4 * nf);
    const bool fv = f < nf;
    __shared__ long long tph_s[9];
    const bool clk = g.dbg != nullptr && blockIdx.x == 0 && tid == 0;
    if (clk) {
        for (int q = 0; q < 8; ++q) tph_s[q] = 0;
        tph_s[8] = wall_clock64();
    }
    auto mark = [&](int ph) {
        if (clk) {
            const long long now = wall_clock64();
            tph_s[ph] += now - tph_s[8];
            tph_s[8] = now;
        }
    };
    constexpr int NI = (RW * 128 + LH_THREADS - 1) / LH_THREADS;  // (row, longitude) items per thread (n <= 126)
    d2 dacc[LH_AK];
#pragma unroll
    for (int u = 0; u < LH_AK; ++u) dacc[u] = d2{0.0, 0.0};
    double pp = 0.0;
    const int qn = (n + 3) / 4;  // longitudes per quarter of the forward transform
    bool first = true;
    for (int base = blockIdx.x; base < g.nc; base += gridDim.x * RW) {
        // ---- requests: the rows' table, R^, the twiddles, the operands of the rows' updates
        d2 th[RW][LH_AK];
#pragma unroll
        for (int r = 0; r < RW; ++r) {
            const int c = base + r * gridDim.x;
            const d2 *Tg = g.That + (int64_t)(c < g.nc ? c : base) * na * nf;
#pragma unroll
            for (int u = 0; u < LH_AK; ++u) {
                const int aa = ag * LH_AK + u;
                th[r][u] = (fv && aa < na && c < g.nc) ? Tg[aa * nf + f] : d2{0.0, 0.0};
            }
        }
        if (first) {
            for (int e = tid; e < n; e += LH_THREADS) tws[e] = g.tw[e];
            if (mode & SW_ADJ) {
                const int tot = na * nf;
                for (int e0 = 0; e0 < tot; e0 += 8 * LH_THREADS) {
                    d2 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u * LH_THREADS + tid;
                        v[u] = g.Rhat[e < tot ? e : tot - 1];
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = e0 + u * LH_THREADS + tid;
                        if (e < tot) Rh[e] = v[u];
                    }
                }
            }
            first = false;
        }
        // items (row r, longitude k) of this thread: it = tid + q * LH_THREADS -> r = it / n, k = it % n
        int64_t ij[NI];
        bool iv[NI];
        double u_w[NI], u_x[NI], u_g[NI], u_p[NI], u_pn[NI], u_hi[NI], u_lo[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int it = tid + q * LH_THREADS, r = it / n, k = it - r * n;
            const int c = base + r * gridDim.x;
            iv[q] = r < RW && c < g.nc;
            ij[q] = (int64_t)(iv[q] ? c : base) * n + k;
            u_w[q] = 1.0;
            u_x[q] = u_g[q] = u_p[q] = u_pn[q] = u_hi[q] = u_lo[q] = 0.0;
            if (iv[q]) {
                const int64_t j = ij[q];
                u_w[q] = wm ? wm[j] : 1.0;
                u_x[q] = (mode & (SW_UPD | SW_FWD)) ? a.x_in[j] : 0.0;
                if (mode & SW_ADJ) {
                    u_g[q] = a.greg ? a.greg[j] : 0.0;
                    if (mode & (SW_PFIN | SW_UPD)) u_p[q] = a.p_in[j];
                    if (mode & SW_SPEC) u_pn[q] = a.pn_in[j];
                    if (mode & SW_UPD) {
                        u_hi[q] = a.high[j];
                        u_lo[q] = a.low[j];
                    }
                }
            }
        }
        __syncthreads();  // Rh, tws in place (first group); the previous group is done with Gp / xs / Xp
        mark(0);
        if (mode & SW_ADJ) {
            // S^_r[f] = sum_a conj(T^_r[a][f]) R^[a][f]: this thread's classes, all rows from one read of R^
            if (fv) {
                d2 rr[LH_AK];
#pragma unroll
                for (int u = 0; u < LH_AK; ++u) {
                    const int aa = ag * LH_AK + u;
                    rr[u] = Rh[(aa < na ? aa : na - 1) * nf + f];  // (classes past the last: th is zero there)
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
            mark(1);
            __syncthreads();
            for (int e = tid; e < RW * nf; e += LH_THREADS) {
                const int r = e / nf, ff = e - r * nf;
                d2 s = Gp[(r * 4) * nf + ff];
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    s.x += Gp[(r * 4 + q) * nf + ff].x;
                    s.y += Gp[(r * 4 + q) * nf + ff].y;
                }
                // weight of the frequency in the inverse transform of a real sequence
                const double wf = (ff == 0 || (2 * ff == n)) ? 1.0 : 2.0;
                Gh[e] = d2{s.x * wf, s.y * wf};
            }
            __syncthreads();
            mark(2);
        }
        // s[k] = (1 / n) sum_f w_f Re(S^[f] e^{+2 pi i f k / n}), gradient, update (hmc.py:114-152)
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int it = tid + q * LH_THREADS, r = it / n, k = it - r * n;
            double xj = u_x[q];
            const double iwj = (u_w[q] != 0.0) ? 1.0 / u_w[q] : 1.0;
            if ((mode & SW_ADJ) && iv[q]) {
                const int64_t j = ij[q];
                const double s = lh_idft_part(Gh + r * nf, tws, 0, nf, k, n);
                const double t = (s / (double)n) * iwj;
                const double grad = 2.0 * t + u_g[q];
                if (mode & SW_GOUT) a.g_out[j] = grad;
                if (mode & SW_PFIN) {
                    const double pf = u_p[q] - a.c_p * grad;
                    pp += pf * pf;
                    if (!(mode & SW_SPEC)) a.p_out[j] = pf;
                }
                if (mode & SW_UPD) {
                    const double psrc = (mode & SW_SPEC) ? u_pn[q] : u_p[q];
                    double pj = psrc - a.c_u * grad;
                    xj = xj + a.dt * pj;
                    if (xj > u_hi[q]) {
                        xj = u_hi[q];
                        pj = -pj;
                    } else if (xj < u_lo[q]) {
                        xj = u_lo[q];
                        pj = -pj;
                    }
                    a.p_out[j] = pj;
                    a.x_out[j] = xj;
                }
            }
            if ((mode & SW_FWD) && r < RW) xs[r * n + k] = iv[q] ? xj * iwj : 0.0;
        }
        mark(3);
        if (mode & SW_FWD) {
            __syncthreads();
            // X^_r[f] = sum_k xs_r[k] e^{-2 pi i f k / n}: quarter ag of the longitudes, then the four quarters
            if (fv) {
                const int k0 = ag * qn, k1 = (k0 + qn < n) ? k0 + qn : n;
#pragma unroll
                for (int r = 0; r < RW; ++r) Xp[(r * 4 + ag) * nf + f] = lh_dft_part(xs + r * n, tws, k0 < n ? k0 : n, k1, f, n);
            }
            mark(4);
            __syncthreads();
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
        }
        mark(5);
    }
    if (mode & SW_PFIN) {
        const double t = block_allreduce_sum(pp, red, LH_THREADS / 64);
        if (tid == 0) a.pp_part[blockIdx.x] = t;
    }
    if ((mode & SW_FWD) && fv) {
        d2 *out = g.Dpart + (int64_t)blockIdx.x * na * nf;
#pragma unroll
        for (int u = 0; u < LH_AK; ++u) {
            const int aa = ag * LH_AK + u;
            if (aa < na) out[aa * nf + f] = dacc[u];
        }
    }
    mark(7);
    if (clk)
        for (int q = 0; q < 8; ++q) g.dbg[q] += tph_s[q];
}

// D^ = sum of the workgroups' partials, inverse transform per class, scatter to the observations of the
// class (slab row 0), per-class sum of the predicted data (dsum[a]).  Block = class a.
__global__ void __launch_bounds__(512) lonsymh_post_kernel(LonHarmGeom g, int nparts, int64_t ld, double *__restrict__ out,
                                                           double *__restrict__ dsum)
{
    __shared__ d2 Dp[8][64];
    __shared__ d2 Dh[64];
    __shared__ d2 tws[1024];
    __shared__ double Sp[4][1024];
    __shared__ double red[8];
    const int a = blockIdx.x, tid = threadIdx.x, f = tid & 63, q = tid >> 6;
    const int n = g.n, nf = g.nf, na = g.na;
    for (int m = tid; m < n; m += 512) tws[m] = g.tw[m];
    if (f < nf) {
        // parts q, q + 8, ...: eight loads in flight
        d2 s = d2{0.0, 0.0};
        const d2 *src = g.Dpart + (int64_t)a * nf + f;
        const int64_t stride = (int64_t)na * nf;
        int w = q;
        for (; w + 56 < nparts; w += 64) {
            d2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = src[(int64_t)(w + 8 * u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                s.x += v[u].x;
                s.y += v[u].y;
            }
        }
        for (; w < nparts; w += 8) {
            const d2 v = src[(int64_t)w * stride];
            s.x += v.x;
            s.y += v.y;
        }
        Dp[q][f] = s;
    }
    __syncthreads();
    if (tid < nf) {
        d2 s = Dp[0][tid];
#pragma unroll
        for (int u = 1; u < 8; ++u) {
            s.x += Dp[u][tid].x;
            s.y += Dp[u][tid].y;
        }
        const double wf = (tid == 0 || (2 * tid == n)) ? 1.0 : 2.0;
        Dh[tid] = d2{s.x * wf, s.y * wf};
    }
    __syncthreads();
    // inverse transform: the frequencies in np ranges, one per group of n threads
    const int np = (512 / n) < 1 ? 1 : ((512 / n) > 4 ? 4 : 512 / n), fpp = (nf + np - 1) / np;
    for (int m0 = 0; m0 < n; m0 += 512) {  // (one trip while n <= 512)
        const int part = np > 1 ? tid / n : 0, m = np > 1 ? tid - part * n : m0 + tid;
        if (part < np && m < n) {
            const int f0 = part * fpp, f1 = (f0 + fpp < nf) ? f0 + fpp : nf;
            Sp[part][m] = lh_idft_part(Dh, tws, f0 < nf ? f0 : nf, f1, m, n);
        }
    }
    __syncthreads();
    double rs = 0.0;
    for (int m = tid; m < n; m += 512) {
        double s = Sp[0][m];
        for (int u = 1; u < np; ++u) s += Sp[u][m];
        const double d = s / (double)n;
        const int e = a * n + m, i0 = g.slot_first[e];
        if (i0 >= 0) {
            out[i0] = d;
            rs += d;
            const int x = g.slot_x[e];
            if (x >= 0)
                for (int qq = g.xptr[x]; qq < g.xptr[x + 1]; ++qq) {
                    out[g.xobs[qq]] = d;
                    rs += d;
                }
        }
    }
    if (a == 0)
        for (int64_t i = g.N + tid; i < ld; i += 512) out[i] = 0.0;  // (the padding rows of the slab row)
    const double t = block_allreduce_sum(rs, red, 8);
    if (tid == 0 && dsum) dsum[a] = t;
}

// ---- the epilogue of an evaluation in ONE launch behind the sweep: what lonsymh_post_kernel, the one-launch
// epilogue of kernels.hip.h (reduce_finish_kernel) and the NEXT pass's lonsymh_rhat_kernel do in three.
// Blocks [0, na): class a -- sum of the sweep's D^ partials, inverse transform, the class's predicted data; the
// mean of d + grav_fix needs every class's sum: the classes exchange them as tagged granule pairs (the data is
// the flag; the na class blocks are the first of the grid, 256 threads each: resident together); residuals of
// the class's observations (potential.py:700-706), their |r|^2 share, and R^[a] for the next adjoint.
// Blocks [na, na + n_regpart): the regulariser of 256 cells each (reg_block, as in reduce_finish_kernel).
struct LhEpiArgs {
    int nparts, n_dpart;
    int64_t ld, N;
    double gfix_sum;
    const double *gfix, *dobs_c;
    double *d, *r, *scal;
    double *r2part;      // n_dpart entries (class a writes entry a, the rest are zero)
    unsigned long long *csum;  // na granule pairs: the classes' sums of d
    unsigned tag;
    unsigned *abort_w;
    RegArgs ra;
};

__global__ void __launch_bounds__(256) lonsymh_epilogue_kernel(LonHarmGeom g, LhEpiArgs e)
{
    __shared__ d2 Dp[4][64];
    __shared__ d2 Dh[64];
    __shared__ d2 tws[1024];
    __shared__ double Sp[2][1024];
    __shared__ double row[1024];
    __shared__ double red[8];
    __shared__ int ok_s;
    const int tid = threadIdx.x;
    const int n = g.n, nf = g.nf, na = g.na;
    if ((int)blockIdx.x >= na) {
        reg_block(e.ra, blockIdx.x - na, red);
        return;
    }
    const int a = blockIdx.x, f = tid & 63, q = tid >> 6;
    for (int m = tid; m < n; m += 256) tws[m] = g.tw[m];
    if (tid == 0) ok_s = 1;
    if (f < nf) {
        // parts q, q + 4, ...: ten loads in flight
        d2 s = d2{0.0, 0.0};
        const d2 *src = g.Dpart + (int64_t)a * nf + f;
        const int64_t stride = (int64_t)na * nf;
        int w = q;
        for (; w + 36 < e.nparts; w += 40) {
            d2 v[10];
#pragma unroll
            for (int u = 0; u < 10; ++u) v[u] = src[(int64_t)(w + 4 * u) * stride];
#pragma unroll
            for (int u = 0; u < 10; ++u) {
                s.x += v[u].x;
                s.y += v[u].y;
            }
        }
        for (; w < e.nparts; w += 4) {
            const d2 v = src[(int64_t)w * stride];
            s.x += v.x;
            s.y += v.y;
        }
        Dp[q][f] = s;
    }
    __syncthreads();
    if (tid < nf) {
        d2 s = Dp[0][tid];
#pragma unroll
        for (int u = 1; u < 4; ++u) {
            s.x += Dp[u][tid].x;
            s.y += Dp[u][tid].y;
        }
        const double wf = (tid == 0 || (2 * tid == n)) ? 1.0 : 2.0;
        Dh[tid] = d2{s.x * wf, s.y * wf};
    }
    __syncthreads();
    // inverse transform: two frequency ranges (n <= 126: two groups of n threads)
    {
        const int part = tid / n, m = tid - part * n, fpp = (nf + 1) / 2;
        if (part < 2) {
            const int f0 = part * fpp, f1 = (f0 + fpp < nf) ? f0 + fpp : nf;
            Sp[part][m] = lh_idft_part(Dh, tws, f0 < nf ? f0 : nf, f1, m, n);
        }
    }
    __syncthreads();
    // the class's predicted data at its slots; the class's sum of d (+ grav_fix) over its observations
    double dm = 0.0, rs = 0.0;
    int i0 = -1;
    if (tid < n) {
        dm = (Sp[0][tid] + Sp[1][tid]) / (double)n;
        const int sl = a * n + tid;
        i0 = g.slot_first[sl];
        if (i0 >= 0) {
            rs += dm + (e.gfix ? e.gfix[i0] : 0.0);
            const int x = g.slot_x[sl];
            if (x >= 0)
                for (int qq = g.xptr[x]; qq < g.xptr[x + 1]; ++qq) rs += dm + (e.gfix ? e.gfix[g.xobs[qq]] : 0.0);
        }
    }
    const double csum = block_allreduce_sum(rs, red, 4);
    if (tid == 0) st_gran(e.csum + 2 * a, e.tag, csum);
    // every class's sum -> the mean (summed in class order: identical bits in every block)
    if (tid < 64) {
        double tot = 0.0;
        bool got = true;
        for (int a0 = 0; a0 < na && got; a0 += 64) {
            double v = 0.0;
            const int aa = a0 + tid;
            got = res_poll(e.abort_w, [&]() -> bool { return aa >= na || ld_gran(e.csum + 2 * aa, e.tag, v); });
            // (lane order inside the group of 64, groups in order)
            tot += wave_sum_dpp(v);
        }
        if (tid == 0) {
            red[7] = tot;
            if (!got) ok_s = 0;
        }
    }
    __syncthreads();
    if (ok_s == 0) return;  // (timed out: the host sees the abort word)
    const double mean = red[7] / (double)e.N;
    // residuals of the class's observations, R[a][m] = their sum per slot, |r|^2 share
    double r2 = 0.0;
    if (tid < n) {
        double Rm = 0.0;
        if (i0 >= 0) {
            const int sl = a * n + tid;
            auto one = [&](int i) {
                const double dinv = dm + (e.gfix ? e.gfix[i] : 0.0);
                const double ri = (dinv - mean) - e.dobs_c[i];
                e.d[i] = dm;
                e.r[i] = ri;
                Rm += ri;
                r2 += ri * ri;
            };
            one(i0);
            const int x = g.slot_x[sl];
            if (x >= 0)
                for (int qq = g.xptr[x]; qq < g.xptr[x + 1]; ++qq) one(g.xobs[qq]);
        }
        row[tid] = Rm;
    }
    const double r2c = block_allreduce_sum(r2, red, 4);
    if (tid == 0) {
        e.r2part[a] = r2c;
        if (a == 0) e.scal[3] = mean;
    }
    if (a == 0) {
        for (int t = na + tid; t < e.n_dpart; t += 256) e.r2part[t] = 0.0;
        for (int64_t i = e.N + tid; i < e.ld; i += 256) {  // (the padding rows)
            e.d[i] = 0.0;
            e.r[i] = 0.0;
        }
    }
    __syncthreads();  // row complete
    // R^[a][f] for the next adjoint pass
    {
        const int qn = (n + 3) / 4;
        if (f < nf) {
            const int k0 = q * qn, k1 = (k0 + qn < n) ? k0 + qn : n;
            Dp[q][f] = lh_dft_part(row, tws, k0 < n ? k0 : n, k1, f, n);
        }
        __syncthreads();
        if (tid < nf) {
            d2 sres = Dp[0][tid];
#pragma unroll
            for (int u = 1; u < 4; ++u) {
                sres.x += Dp[u][tid].x;
                sres.y += Dp[u][tid].y;
            }
            g.Rhat[a * nf + tid] = sres;
        }
    }
}

}  // namespace ghk
