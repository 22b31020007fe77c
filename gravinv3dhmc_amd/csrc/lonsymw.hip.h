// The longitude-harmonic shift-invariant store for grids BEYOND the register form of lonsymh.hip.h (gfx950):
// more than 126 longitudes (nf = n / 2 + 1 > 64) or more than 64 observation classes -- a 1-degree global grid has
// n = 360, 181 classes, 1 800 cell rows: T^ is 0.94 GB, a cell row's slab of it (n_a nf complex = 524 KB) fits no
// workgroup's registers, R^ (same size) no LDS.  Same arithmetic as lonsymh.hip.h
//     S^[c][f]  = sum_a conj(T^[c][a][f]) R^[a][f]          adjoint
//     D^[a][f]  = sum_c T^[c][a][f] X^[c][f]                forward
// as STREAMING passes over T^ -- an HBM-bound kernel pair, every load coalesced along f:
//   lonsymw_rhat_kernel     R^ of the residual (block = class), as lonsymh_rhat_kernel without the 64-frequency limit
//   lonsymw_sweep_kernel    block = cell row: S^ from the row's slab of T^ (read once, R^ from L2), inverse
//                           transform, gradient + leapfrog update (hmc.py:114-152), transform of the new positions ->
//                           X^[c][f] (n_c nf complex in memory: 5 MB on the 1-degree grid)
//   lonsymw_forward_kernel  thread = (a, f) -- the flattened index of a row of T^, no idle lanes --, sum over a RANGE of
//                           cell rows: D^ partials [parts][n_a][nf]
//   lonsymw_post_kernel     sum of the parts, inverse transform per class, scatter to the class's observations
// T^ is read twice per leapfrog step (the forward product needs every row's X^, which needs every row's update): 2 x 0.94 GB
// on the 1-degree grid = 0.24 ms at 8 TB/s, where the dense kernel of that grid (65 341 x 648 000 doubles = 339 GB) fits no
// GPU.  On a grid symmetric about the equator ONE row of T^ stands for a north-south mirrored pair of cell rows (the
// struct's item_* fields): half the table, every entry read serves both rows.  No inter-workgroup waits: four plain launches per step.  Sums in a fixed order: reproducible bit for bit.
// Reference arithmetic: gravmag/_tesseroid_numba.py:207-222 (cos(lon - lon')), gravmag/tesseroid.py:189-232,
// inversion/potential.py:698,708, inversion/hmc.py:114-152; geometry family example/global/SetPMTS.txt.
#pragma once
#include "lonsymh.hip.h"

namespace ghk {

struct LonWideGeom {
    int n, nf, nfp, na, nc, parts, rows_per_part;
    int brk;                 // diagnostic (GRAVHMC_LW_BREAK): 1 no stream of T^, 2 no inverse transform, 4 no transform of xs
    const d2 *That;          // [nc][na][nfp]   (nfp = nf rounded up to 8 complex: rows start on 128-byte lines; the pad is zero)
    const d2 *tw;            // n
    d2 *Rhat;                // [na][nfp]
    d2 *Xhat;                // [nc][nf]
    d2 *Dpart;               // [parts][na][nfp]
    const int *slot_first;   // na * n
    const int *slot_x;       // na * n: the slot's entry in xptr, or -1
    const int *xptr, *xobs;
    int64_t N;
    // North-south mirror (grids symmetric about the equator: K of (cell row c, class a) = K of (mirror row, mirror class)):
    // T^ holds ONE row per pair of mirrored cell rows -- item p stands for the rows item_c[p] and item_c2[p] (-1: the row
    // is its own mirror) -- and every entry read serves both.  nitems rows in That; planes = 2 planes of D^ partials
    // (the second one lands on the mirrored class).  item_c == nullptr: no mirror, item p = cell row p.
    int nitems, planes;
    const int *item_c, *item_c2, *amir;
};

constexpr int LW_THREADS = 256;
constexpr int LW_NMAX = 1024;  // longitudes per cell row (the transforms' tables live in LDS)
constexpr int LW_NP = (LW_NMAX / 2 + 1 + LW_THREADS - 1) / LW_THREADS;  // pairs of longitudes (m, n - m) per thread at most

static inline size_t lonsymw_lds_doubles(int n, int nf)
{
    return 2 * (size_t)n + 16 * (size_t)nf + 4 * (size_t)nf + 2 * (size_t)nf + 16;
}

// The two transforms of a cell row from the pairs of longitudes (m, n - m), m = 0 .. n / 2 (half the terms: the
// twiddle reads -- a different LDS address per lane -- are what the transforms cost):
//   s[m] = E - O, s[n - m] = E + O   with E = sum_f Re(H[f]) cos(2 pi f m / n), O = sum_f Im(H[f]) sin(2 pi f m / n)
//   X^[f] = sum_m (xe[m] cos(2 pi f m / n), -xo[m] sin(2 pi f m / n)),  xe = x[m] + x[n - m], xo = x[m] - x[n - m]
//   (m = 0 and, n even, m = n / 2 stand alone: xe = x[m], xo = 0)
__device__ __forceinline__ void lw_idft_pair(const d2 *H, const d2 *tws, int nf, int m, int n, double &E, double &O)
{
    double e = 0.0, o = 0.0;
    int idx = 0, ff = 0;
    for (; ff + 8 <= nf; ff += 8) {
        d2 h[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            h[u] = H[ff + u];
            w[u] = tws[idx];
            idx += m;
            if (idx >= n) idx -= n;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            e += h[u].x * w[u].x;
            o += h[u].y * w[u].y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; ff < nf; ++ff) {
        const d2 h = H[ff], w = tws[idx];
        e += h.x * w.x;
        o += h.y * w.y;
        idx += m;
        if (idx >= n) idx -= n;
    }
    E = e;
    O = o;
}

__device__ __forceinline__ d2 lw_dft_pairs(const d2 *XE, const d2 *tws, int np, int f, int n)
{
    d2 acc = d2{0.0, 0.0};
    int idx = 0, m = 0;
    for (; m + 8 <= np; m += 8) {
        d2 x[8], w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            x[u] = XE[m + u];
            w[u] = tws[idx];
            idx += f;
            if (idx >= n) idx -= n;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            acc.x += x[u].x * w[u].x;
            acc.y -= x[u].y * w[u].y;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    for (; m < np; ++m) {
        const d2 x = XE[m], w = tws[idx];
        acc.x += x.x * w.x;
        acc.y -= x.y * w.y;
        idx += f;
        if (idx >= n) idx -= n;
    }
    return acc;
}

__global__ void __launch_bounds__(LW_THREADS) lonsymw_rhat_kernel(LonWideGeom g, const double *__restrict__ r)
{
    __shared__ double row[LW_NMAX];
    __shared__ d2 tws[LW_NMAX];
    const int a = blockIdx.x, n = g.n, nf = g.nf, tid = threadIdx.x;
    for (int m = tid; m < n; m += LW_THREADS) {
        const int e = a * n + m, idx = g.slot_first[e];
        double v = idx >= 0 ? r[idx] : 0.0;
        const int x = g.slot_x[e];
        if (x >= 0)
            for (int q = g.xptr[x]; q < g.xptr[x + 1]; ++q) v += r[g.xobs[q]];
        row[m] = v;
        tws[m] = g.tw[m];
    }
    __syncthreads();
    for (int f = tid; f < nf; f += LW_THREADS) g.Rhat[(int64_t)a * g.nfp + f] = lh_dft_part(row, tws, 0, n, f, n);
}

// modes of SweepArgs as lonsym_sweep_kernel / lonsymh_sweep_kernel; SW_FWD leaves X^ of every cell row in g.Xhat.
// NP = pairs of longitudes per thread (nf <= NP * 256): the update's operands of a thread's pairs wait in registers
// while the row's slab of T^ streams -- with NP = 3 for every grid the kernel took 189 registers (two waves per SIMD);
// NP = 1 (n <= 510): 127, four waves per SIMD = four workgroups per CU.
template <int NP>
__device__ __forceinline__ void lonsymw_sweep_body(const LonWideGeom &g, const SweepArgs &a, const double *__restrict__ wm)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = g.n, nf = g.nf, nfp = g.nfp, na = g.na;
    const int mode = a.mode;
    d2 *tws = reinterpret_cast<d2 *>(smem);                   // n
    d2 *Gp = tws + n;                                         // 2 x 4 x nf: partial S^ of the waves' classes, both rows of the item
    d2 *Gh = Gp + 8 * nf;                                     // 2 x nf
    d2 *XE = Gh + 2 * nf;                                     // nf: (even, odd) parts of xs of the pairs (m, n - m)
    double *red = reinterpret_cast<double *>(XE + nf);
    for (int e = tid; e < n; e += LW_THREADS) tws[e] = g.tw[e];
    double pp = 0.0;
    for (int p = blockIdx.x; p < g.nitems; p += gridDim.x) {
        const int c1 = g.item_c ? g.item_c[p] : p, c2 = g.item_c ? g.item_c2[p] : -1;
        // the operands of a row's updates.  Thread <-> pairs of longitudes m = tid, tid + 256, ...: item 0 = longitude m,
        // item 1 = longitude n - m (none for m = 0 and 2 m = n).  The first row's are requested in front of the stream of T^.
        double u_w[NP][2], u_x[NP][2], u_g[NP][2], u_p[NP][2], u_pn[NP][2], u_hi[NP][2], u_lo[NP][2];
        auto fetch = [&](int c) {
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int m = tid + q * LW_THREADS;
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const int k = w == 0 ? m : n - m;
                    u_w[q][w] = 1.0;
                    u_x[q][w] = u_g[q][w] = u_p[q][w] = u_pn[q][w] = u_hi[q][w] = u_lo[q][w] = 0.0;
                    if (m < nf && (w == 0 || (m > 0 && 2 * m != n))) {
                        const int64_t j = (int64_t)c * n + k;
                        u_w[q][w] = wm ? wm[j] : 1.0;
                        u_x[q][w] = (mode & (SW_UPD | SW_FWD)) ? a.x_in[j] : 0.0;
                        if (mode & SW_ADJ) {
                            u_g[q][w] = a.greg ? a.greg[j] : 0.0;
                            if (mode & (SW_PFIN | SW_UPD)) u_p[q][w] = a.p_in[j];
                            if (mode & SW_SPEC) u_pn[q][w] = a.pn_in[j];
                            if (mode & SW_UPD) {
                                u_hi[q][w] = a.high[j];
                                u_lo[q][w] = a.low[j];
                            }
                        }
                    }
                }
            }
        };
        fetch(c1);
        __syncthreads();  // tws in place; the previous item is done with Gp / Gh / XE
        if (mode & SW_ADJ) {
            // S^[f] = sum_a conj(T^[a][f]) R^[a][f] (and, for the mirrored row, R^ of the mirrored class): wave wv takes the
            // classes wv, wv + 4, ...; eight requests of T^ in flight per wave (four when the entry serves two rows)
            const d2 *Tg = g.That + (int64_t)p * na * nfp;
            for (int f0 = 0; f0 < nf; f0 += 64) {
                const int f = f0 + lane;
                const bool fv = f < nf;
                const int fc = fv ? f : nf - 1;
                d2 acc = d2{0.0, 0.0}, acc2 = d2{0.0, 0.0};
                int aa = wv;
                if (!(g.brk & 1)) {
                    if (c2 < 0) {
                        for (; aa + 28 < na; aa += 32) {
                            d2 t[8], rr[8];
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                t[u] = __builtin_nontemporal_load(&Tg[(int64_t)(aa + 4 * u) * nfp + fc]);
                                rr[u] = g.Rhat[(int64_t)(aa + 4 * u) * nfp + fc];
                            }
#pragma unroll
                            for (int u = 0; u < 8; ++u) {
                                acc.x += t[u].x * rr[u].x + t[u].y * rr[u].y;
                                acc.y += t[u].x * rr[u].y - t[u].y * rr[u].x;
                            }
                        }
                        for (; aa < na; aa += 4) {
                            const d2 t = __builtin_nontemporal_load(&Tg[(int64_t)aa * nfp + fc]), rr = g.Rhat[(int64_t)aa * nfp + fc];
                            acc.x += t.x * rr.x + t.y * rr.y;
                            acc.y += t.x * rr.y - t.y * rr.x;
                        }
                    } else {
                        for (; aa + 12 < na; aa += 16) {
                            d2 t[4], rr[4], r2[4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const int ac = aa + 4 * u;
                                t[u] = __builtin_nontemporal_load(&Tg[(int64_t)ac * nfp + fc]);
                                rr[u] = g.Rhat[(int64_t)ac * nfp + fc];
                                r2[u] = g.Rhat[(int64_t)g.amir[ac] * nfp + fc];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                acc.x += t[u].x * rr[u].x + t[u].y * rr[u].y;
                                acc.y += t[u].x * rr[u].y - t[u].y * rr[u].x;
                                acc2.x += t[u].x * r2[u].x + t[u].y * r2[u].y;
                                acc2.y += t[u].x * r2[u].y - t[u].y * r2[u].x;
                            }
                        }
                        for (; aa < na; aa += 4) {
                            const d2 t = __builtin_nontemporal_load(&Tg[(int64_t)aa * nfp + fc]), rr = g.Rhat[(int64_t)aa * nfp + fc],
                                     r2 = g.Rhat[(int64_t)g.amir[aa] * nfp + fc];
                            acc.x += t.x * rr.x + t.y * rr.y;
                            acc.y += t.x * rr.y - t.y * rr.x;
                            acc2.x += t.x * r2.x + t.y * r2.y;
                            acc2.y += t.x * r2.y - t.y * r2.x;
                        }
                    }
                }
                if (fv) {
                    Gp[wv * nf + f] = acc;
                    Gp[(4 + wv) * nf + f] = acc2;
                }
            }
            __syncthreads();
            for (int e = tid; e < 2 * nf; e += LW_THREADS) {
                const int r = e >= nf ? 1 : 0, f = e - r * nf;
                d2 s = Gp[(4 * r) * nf + f];
#pragma unroll
                for (int q = 1; q < 4; ++q) {
                    s.x += Gp[(4 * r + q) * nf + f].x;
                    s.y += Gp[(4 * r + q) * nf + f].y;
                }
                // weight of the frequency in the inverse transform of a real sequence
                const double wf = (f == 0 || (2 * f == n)) ? 1.0 : 2.0;
                Gh[e] = d2{s.x * wf, s.y * wf};
            }
            __syncthreads();
        }
        for (int r = 0; r < 2; ++r) {
            const int c = r == 0 ? c1 : c2;
            if (c < 0) break;  // (uniform over the workgroup)
            if (r == 1) {
                __syncthreads();  // the first row's transform is done with XE
                fetch(c);
            }
            const d2 *Ghr = Gh + r * nf;
            // s[k] = (1 / n) sum_f w_f Re(S^[f] e^{+2 pi i f k / n}), gradient, update (hmc.py:114-152)
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int m = tid + q * LW_THREADS;
                if (m < nf) {
                    double E = 0.0, O = 0.0;
                    if (mode & SW_ADJ) {
                        if (g.brk & 2) {
                            E = Ghr[m].x;
                            O = Ghr[m].y;
                        } else {
                            lw_idft_pair(Ghr, tws, nf, m, n, E, O);
                        }
                    }
                    double xe = 0.0, xo = 0.0;
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        if (w == 1 && (m == 0 || 2 * m == n)) continue;
                        const int k = w == 0 ? m : n - m;
                        const int64_t j = (int64_t)c * n + k;
                        const double iwj = (u_w[q][w] != 0.0) ? 1.0 / u_w[q][w] : 1.0;
                        double xj = u_x[q][w];
                        if (mode & SW_ADJ) {
                            const double s = w == 0 ? E - O : E + O;
                            const double t = (s / (double)n) * iwj;
                            const double grad = 2.0 * t + u_g[q][w];
                            if (mode & SW_GOUT) a.g_out[j] = grad;
                            if (mode & SW_PFIN) {
                                const double pf = u_p[q][w] - a.c_p * grad;
                                pp += pf * pf;
                                if (!(mode & SW_SPEC)) a.p_out[j] = pf;
                            }
                            if (mode & SW_UPD) {
                                const double psrc = (mode & SW_SPEC) ? u_pn[q][w] : u_p[q][w];
                                double pj = psrc - a.c_u * grad;
                                xj = xj + a.dt * pj;
                                if (xj > u_hi[q][w]) {
                                    xj = u_hi[q][w];
                                    pj = -pj;
                                } else if (xj < u_lo[q][w]) {
                                    xj = u_lo[q][w];
                                    pj = -pj;
                                }
                                a.p_out[j] = pj;
                                a.x_out[j] = xj;
                            }
                        }
                        const double xsj = xj * iwj;
                        xe += xsj;
                        xo += w == 0 ? xsj : -xsj;
                    }
                    if (m == 0 || 2 * m == n) xo = 0.0;
                    if (mode & SW_FWD) XE[m] = d2{xe, xo};
                }
            }
            if (mode & SW_FWD) {
                __syncthreads();
                for (int f = tid; f < nf; f += LW_THREADS)
                    g.Xhat[(int64_t)c * nf + f] = (g.brk & 4) ? XE[f] : lw_dft_pairs(XE, tws, nf, f, n);
            }
        }
    }
    if (mode & SW_PFIN) {
        __syncthreads();
        const double t = block_allreduce_sum(pp, red, LW_THREADS / 64);
        if (tid == 0) a.pp_part[blockIdx.x] = t;
    }
}

template <int NP>
__global__ void __launch_bounds__(LW_THREADS) lonsymw_sweep_kernel(LonWideGeom g, SweepArgs a, const double *__restrict__ wm)
{
    lonsymw_sweep_body<NP>(g, a, wm);
}

// D^ partial of the items [part * rows_per_part, ...): thread e = a * nfp + f (one complex of a row of T^); an entry of a
// mirrored pair of cell rows feeds two accumulators (plane 0: class a from the row itself, plane 1: the mirrored class from
// the mirrored row).  UN rows in flight per thread; NT: non-temporal loads of T^.
template <int UN, bool NT>
__global__ void __launch_bounds__(LW_THREADS) lonsymw_forward_kernel(LonWideGeom g)
{
    const int nf = g.nf;
    const int64_t tot = (int64_t)g.na * g.nfp;
    const int64_t e = (int64_t)blockIdx.x * LW_THREADS + threadIdx.x;
    const bool ev = e < tot;
    const int64_t ec = ev ? e : tot - 1;
    const int fp = (int)(ec % g.nfp);
    const int f = fp < nf ? fp : nf - 1;  // (the pad of a row of T^ is zero: any X^ will do)
    const int p0 = blockIdx.y * g.rows_per_part;
    const int p1 = (p0 + g.rows_per_part < g.nitems) ? p0 + g.rows_per_part : g.nitems;
    const bool mir = g.item_c != nullptr;
    d2 acc = d2{0.0, 0.0}, acc2 = d2{0.0, 0.0};
    int p = p0;
    for (; p + UN <= p1; p += UN) {
        d2 t[UN], x[UN], x2[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const d2 *src = &g.That[(int64_t)(p + u) * tot + ec];
            t[u] = NT ? __builtin_nontemporal_load(src) : *src;
            const int c1 = mir ? g.item_c[p + u] : p + u, c2 = mir ? g.item_c2[p + u] : -1;
            x[u] = g.Xhat[(int64_t)c1 * nf + f];
            x2[u] = c2 >= 0 ? g.Xhat[(int64_t)c2 * nf + f] : d2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            acc.x += t[u].x * x[u].x - t[u].y * x[u].y;
            acc.y += t[u].x * x[u].y + t[u].y * x[u].x;
            acc2.x += t[u].x * x2[u].x - t[u].y * x2[u].y;
            acc2.y += t[u].x * x2[u].y + t[u].y * x2[u].x;
        }
    }
    for (; p < p1; ++p) {
        const d2 t = g.That[(int64_t)p * tot + ec];
        const int c1 = mir ? g.item_c[p] : p, c2 = mir ? g.item_c2[p] : -1;
        const d2 x = g.Xhat[(int64_t)c1 * nf + f];
        const d2 x2 = c2 >= 0 ? g.Xhat[(int64_t)c2 * nf + f] : d2{0.0, 0.0};
        acc.x += t.x * x.x - t.y * x.y;
        acc.y += t.x * x.y + t.y * x.x;
        acc2.x += t.x * x2.x - t.y * x2.y;
        acc2.y += t.x * x2.y + t.y * x2.x;
    }
    if (ev) {
        g.Dpart[((int64_t)blockIdx.y * g.planes) * tot + e] = acc;
        if (g.planes > 1) g.Dpart[((int64_t)blockIdx.y * g.planes + 1) * tot + e] = acc2;
    }
}

// block = class a: sum of the parts, inverse transform, scatter to the class's observations (slab row 0), the
// class's sum of predicted data (dsum[a])
__global__ void __launch_bounds__(LW_THREADS) lonsymw_post_kernel(LonWideGeom g, int64_t ld, double *__restrict__ out,
                                                                  double *__restrict__ dsum)
{
    __shared__ d2 Dh[LW_NMAX / 2 + 1];
    __shared__ d2 tws[LW_NMAX];
    __shared__ double red[8];
    const int a = blockIdx.x, tid = threadIdx.x;
    const int n = g.n, nf = g.nf;
    const int64_t tot = (int64_t)g.na * g.nfp;
    for (int m = tid; m < n; m += LW_THREADS) tws[m] = g.tw[m];
    for (int f = tid; f < nf; f += LW_THREADS) {
        d2 s = d2{0.0, 0.0};
        const d2 *src = g.Dpart + (int64_t)a * g.nfp + f;
        for (int p = 0; p < g.parts; ++p) {
            const d2 v = src[(int64_t)p * g.planes * tot];
            s.x += v.x;
            s.y += v.y;
        }
        if (g.planes > 1) {
            // what the mirrored cell rows contribute to this class was accumulated at the mirrored class's entries
            const d2 *src2 = g.Dpart + tot + (int64_t)g.amir[a] * g.nfp + f;
            for (int p = 0; p < g.parts; ++p) {
                const d2 v = src2[(int64_t)p * g.planes * tot];
                s.x += v.x;
                s.y += v.y;
            }
        }
        const double wf = (f == 0 || (2 * f == n)) ? 1.0 : 2.0;
        Dh[f] = d2{s.x * wf, s.y * wf};
    }
    __syncthreads();
    double rs = 0.0;
    for (int m = tid; m < n; m += LW_THREADS) {
        const double d = lh_idft_part(Dh, tws, 0, nf, m, n) / (double)n;
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
        for (int64_t i = g.N + tid; i < ld; i += LW_THREADS) out[i] = 0.0;  // (the padding rows of the slab row)
    const double t = block_allreduce_sum(rs, red, LW_THREADS / 64);
    if (tid == 0 && dsum) dsum[a] = t;
}

}  // namespace ghk
