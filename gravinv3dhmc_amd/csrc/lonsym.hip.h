// Shift-invariant store for regular spherical grids (gfx950): the kernel matrix of a global model is
// never assembled -- and never re-evaluated either.
//
// BASELINE configs[3] (example/global/main_global.py:25-28, SetPMTS.txt) puts a 3-degree tesseroid
// mesh under a 3-degree observation grid.  The gz entry of (observation, tesseroid) depends on the
// two longitudes only through their difference (gravmag/tesseroid.py:189-232,
// _tesseroid_numba.py:207-222: cos(lon - lon')), so with
//     cells       j = c * n + k        c = (layer, latitude band) "cell row", k = 0..n-1 the longitude index,
//                                      w_k = w_0 + k * dlon, n * dlon = 360
//     observations i -> (a_i, m_i)     a = class of equal (latitude, height), lon_i = lon_ref + m_i * dlon
// the whole matrix is  K[i, (c, k)] = T[c][a_i][(m_i - k) mod n]:  n_c x n_a x n numbers (C4: 600 x 61 x
// 120 doubles = 35 MB, L2 / Infinity-Cache resident) instead of N x M (4.25 GB) or 5.3e8 evaluations
// per step.  The table is built once with the reference's own adaptive engine (tess_gz_kernel on the
// synthetic problem "every class at every shift against the cells of longitude index 0": near-field
// pairs included, same subdivision decisions).  Forward and adjoint become circular correlations
// along the longitude, per (class, cell row):
//     S[c][k]  = sum_a sum_m T[c][a][(m - k) mod n] * R[a][m]        R[a][m] = sum of r_i over the slot
//     D[a][m] += sum_k     T[c][a][(m - k) mod n] * xs[c][k]         d_i = D[a_i][m_i]
// The fused leapfrog pass of the dense sweep carries over unchanged: a workgroup owns cell row c,
// holds T[c] (58 KB) and R in LDS, forms the 120 dots, updates the 120 cells (hmc.py:114-152) and adds
// their forward contribution to its D accumulators (registers) -- one read of the 35 MB table per step.
//
// Inner loop (both products): lane = class a, a wave owns 8 consecutive k (adjoint) or m (forward);
// the 8 table values it needs at a step are 8 consecutive shifts, and the next step needs the same
// window moved by one: a circular buffer of 8 registers, ONE new LDS read per 8 FMAs.  Row stride of the
// LDS copies = 1 mod 16 doubles: the lanes' rows fall into different banks.
// Sums run in a fixed order: reproducible bit for bit.  The values differ from the dense engine's by
// the rounding of cos(lon - lon') at a shifted pair of longitudes (~1e-16 of an entry) and by the
// association of the sums; stated tolerance 1e-10, gated by the tests at full size.
#pragma once

namespace ghk {

struct LonSymGeom {
    int n, na, nc;         // longitudes per cell row, observation classes, cell rows
    int SW;                // row stride of the LDS copies (doubles)
    int AG, KB;            // lane groups of 64 classes, blocks of W longitudes: AG * KB work items
    int64_t ldT;           // doubles per cell row of the table
    const double *T;       // [nc][ldT], T[c][a * n + delta]
    const int *slot_first; // na * n: first observation of slot (a, m), or -1
    int n_xslots;          // slots with more than one observation (duplicated longitudes such as +-180)
    const int *xslot;      // n_xslots: their slot numbers
    const int *xptr;       // n_xslots + 1
    const int *xobs;       // their further observations, ascending
    const int *lds_of;     // N: a_i * SW + m_i
    int64_t N;
    long long *dbg;        // optional: 8 accumulated phase times (100 MHz ticks) of workgroup 0, thread 0
};

constexpr int LS_THREADS = 1024;
constexpr int LS_WAVES = 16;
constexpr int LS_MAXITEMS = 4;  // work items per wave at most (8 forward accumulators each)

// acc[u] += sum over t of window(t)[u] * v(t): see the header.  ADJ: u = 7 - j and v = Rrow[t] (this
// lane's row of R); forward: u = j and v = xsr[t] (uniform).  Trow = this lane's row of T[c]; b0 = first
// shift of the window at t = 0; steps = multiple of 8 (v is zero beyond n).
template <bool ADJ, int W>
__device__ __forceinline__ void ls_correlate(double (&acc)[W], const double *Trow, const double *v, int b0, int n, int steps)
{
    double w[W];
    int ti = b0;
#pragma unroll
    for (int x = 0; x < W; ++x) {
        w[x] = Trow[ti];
        ti = ti + 1 == n ? 0 : ti + 1;
    }
    for (int t0 = 0; t0 < steps; t0 += W) {
#pragma unroll
        for (int t = 0; t < W; ++t) {
            const double vv = v[t0 + t];
            const double nx = Trow[ti];  // the element that joins the window at the next step
            ti = ti + 1 == n ? 0 : ti + 1;
#pragma unroll
            for (int j = 0; j < W; ++j) acc[ADJ ? W - 1 - j : j] = fma(w[(t + j) & (W - 1)], vv, acc[ADJ ? W - 1 - j : j]);
            w[t] = nx;
        }
    }
}

// The fused pass (modes of SweepArgs as sweep_kernel / mf_fused_kernel).  ITEMS = work items per wave
// (AG * KB <= 16 ITEMS): the forward accumulators are registers.  W = longitudes a work item owns
// (8 or 16): every table value read from LDS feeds W FMAs -- at 8 the two LDS reads per step of the
// 15 waves (1 KB per 512 FMAs) saturate the LDS before the VALU (measured: VALU busy 43 %); at 16 half
// as many waves do twice the FMAs per read.
// T = threads per workgroup (1024, or 512 when eight waves cover the work items: twice the registers).
template <int ITEMS, int W, int T>
__global__ void __launch_bounds__(T)
lonsym_sweep_kernel(LonSymGeom g, SweepArgs a, const double *__restrict__ wm)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    constexpr int NWV = T / 64;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = g.n, na = g.na, SW = g.SW;
    const int mode = a.mode;
    const int steps = (n + W - 1) / W * W;
    double *Tc = smem;                                   // na x SW
    double *Rg = Tc + (size_t)na * SW;                   // (na + 1) x SW: row na is zero
    double *xsr = Rg + (size_t)(na + 1) * SW;            // steps + 8: xs reversed, zero beyond n
    double *Sp = xsr + steps + 8;                        // AG x (KB * W)
    double *red = Sp + (size_t)g.AG * g.KB * W;          // 32
    const int nitems = g.AG * g.KB;
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

    // A thread's share of a cell row's table on its way from memory to LDS: requested while the
    // previous row is still being worked on, parked after that row's last barrier.
    constexpr int TPER = 8192 / T;  // T threads x TPER >= na * n for everything that fits the LDS
    double tnext[TPER];
    auto t_fetch = [&](int c) {
        const double *Tg = g.T + (int64_t)c * g.ldT;
        const int tot = na * n;
#pragma unroll
        for (int q = 0; q < TPER; ++q) {
            const int e = tid + q * T;
            tnext[q] = Tg[e < tot ? e : tot - 1];
        }
    };
    auto t_park = [&]() {
        const int tot = na * n;
#pragma unroll
        for (int q = 0; q < TPER; ++q) {
            const int e = tid + q * T;
            if (e < tot) {
                const int aa = e / n;
                Tc[aa * SW + (e - aa * n)] = tnext[q];
            }
        }
    };
    if ((int)blockIdx.x < g.nc) t_fetch(blockIdx.x);  // (in flight behind the gather of R below)
    // R[a][m] = sum of r over the slot's observations; zero elsewhere (padding, the extra row)
    for (int e = tid; e < (na + 1) * SW; e += T) Rg[e] = 0.0;
    for (int e = tid; e < steps + 8; e += T) xsr[e] = 0.0;
    __syncthreads();
    if (mode & SW_ADJ) {
        // the first observation of every slot in batches of eight (index, then value: two round trips
        // per batch -- slot by slot, pointer -> index -> value were three per slot, 16 slots per thread
        // one after the other: a third of the pass), then the rare further observations of a slot
        const int tot = na * n;
        for (int e0 = 0; e0 < tot; e0 += 8 * T) {
            int idx[8];
            double val[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q * T + tid;
                idx[q] = e < tot ? g.slot_first[e] : -1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) val[q] = idx[q] >= 0 ? a.r[idx[q]] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q * T + tid;
                if (e < tot) {
                    const int aa = e / n;
                    Rg[aa * SW + (e - aa * n)] = val[q];
                }
            }
        }
        __syncthreads();
        for (int x = tid; x < g.n_xslots; x += T) {
            const int e = g.xslot[x], aa = e / n;
            double t = Rg[aa * SW + (e - aa * n)];
            for (int q = g.xptr[x]; q < g.xptr[x + 1]; ++q) t += a.r[g.xobs[q]];
            Rg[aa * SW + (e - aa * n)] = t;
        }
    }
    double dacc[ITEMS][W];
#pragma unroll
    for (int q = 0; q < ITEMS; ++q)
#pragma unroll
        for (int u = 0; u < W; ++u) dacc[q][u] = 0.0;
    double pp = 0.0;
    mark(0);
    for (int c = blockIdx.x; c < g.nc; c += gridDim.x) {
        __syncthreads();  // (the previous row's forward has finished with Tc and xsr; first trip: Rg is complete)
        mark(6);
        t_park();
        // the operands of this row's update (thread tid < n: cell (c, tid)): requested now, looked at
        // after the dots -- their round trip used to stand between the dots and the forward
        const int64_t j = (int64_t)c * n + tid;
        double u_w = 1.0, u_x = 0.0, u_g = 0.0, u_p = 0.0, u_pn = 0.0, u_hi = 0.0, u_lo = 0.0;
        if (tid < n) {
            u_w = wm ? wm[j] : 1.0;
            u_x = (mode & (SW_UPD | SW_FWD)) ? a.x_in[j] : 0.0;
            if (mode & SW_ADJ) {
                u_g = a.greg ? a.greg[j] : 0.0;
                if (mode & (SW_PFIN | SW_UPD)) u_p = a.p_in[j];
                if (mode & SW_SPEC) u_pn = a.pn_in[j];
                if (mode & SW_UPD) {
                    u_hi = a.high[j];
                    u_lo = a.low[j];
                }
            }
        }
        __syncthreads();
        mark(1);
        double xj = u_x, iwj = 1.0;
        if (mode & SW_ADJ) {
#pragma unroll 1
            for (int it = wave; it < nitems; it += NWV) {
                const int ag = it / g.KB, kb = it - ag * g.KB;
                const int aa = ag * 64 + lane;
                const double *Trow = Tc + (aa < na ? aa : na - 1) * SW;
                const double *Rrow = Rg + (aa < na ? aa : na) * SW;  // row na: zeros
                int b0 = (-(kb * W + W - 1)) % n;
                if (b0 < 0) b0 += n;
                double acc[W];
#pragma unroll
                for (int u = 0; u < W; ++u) acc[u] = 0.0;
                ls_correlate<true, W>(acc, Trow, Rrow, b0, n, steps);
#pragma unroll
                for (int u = 0; u < W; ++u) {
                    const double s = wave_sum_dpp(acc[u]);
                    if (lane == 0) Sp[ag * g.KB * W + kb * W + u] = s;
                }
            }
            mark(2);
            __syncthreads();
            mark(7);
        }
        if (tid < n) iwj = (u_w != 0.0) ? 1.0 / u_w : 1.0;
        if ((mode & SW_ADJ) && tid < n) {
            double t = 0.0;
            for (int ag = 0; ag < g.AG; ++ag) t += Sp[ag * g.KB * W + tid];
            t = t * iwj;
            const double grad = 2.0 * t + u_g;
            if (mode & SW_GOUT) a.g_out[j] = grad;
            if (mode & SW_PFIN) {
                const double pf = u_p - a.c_p * grad;
                pp += pf * pf;
                if (!(mode & SW_SPEC)) a.p_out[j] = pf;
            }
            if (mode & SW_UPD) {
                const double psrc = (mode & SW_SPEC) ? u_pn : u_p;
                double pj = psrc - a.c_u * grad;
                xj = xj + a.dt * pj;
                if (xj > u_hi) {
                    xj = u_hi;
                    pj = -pj;
                } else if (xj < u_lo) {
                    xj = u_lo;
                    pj = -pj;
                }
                a.p_out[j] = pj;
                a.x_out[j] = xj;
            }
        }
        mark(3);
        // the next row's table: in flight behind the forward below
        if (c + (int)gridDim.x < g.nc) t_fetch(c + gridDim.x);
        if (mode & SW_FWD) {
            if (tid < n) xsr[n - 1 - tid] = xj * iwj;
            __syncthreads();
#pragma unroll
            for (int qq = 0; qq < ITEMS; ++qq) {
                const int it = wave + qq * NWV;
                if (it < nitems) {
                    const int ag = it / g.KB, mb = it - ag * g.KB;
                    const int aa = ag * 64 + lane;
                    const double *Trow = Tc + (aa < na ? aa : na - 1) * SW;
                    const int b0 = (mb * W + 1) % n;
                    ls_correlate<false, W>(dacc[qq], Trow, xsr, b0, n, steps);
                }
            }
        }
        mark(4);
    }
    if ((mode & SW_PFIN)) {
        // sum of p^2 over this workgroup's cells: threads tid < n hold parts
        const double t = block_allreduce_sum(tid < n ? pp : 0.0, red, NWV);
        if (tid == 0) a.pp_part[blockIdx.x] = t;
    }
    if (mode & SW_FWD) {
        __syncthreads();  // everybody is done with Tc: it takes D[a][m]
#pragma unroll
        for (int qq = 0; qq < ITEMS; ++qq) {
            const int it = wave + qq * NWV;
            if (it < nitems) {
                const int ag = it / g.KB, mb = it - ag * g.KB;
                const int aa = ag * 64 + lane;
                if (aa < na) {
#pragma unroll
                    for (int u = 0; u < W; ++u)
                        if (mb * W + u < n) Tc[aa * SW + mb * W + u] = dacc[qq][u];
                }
            }
        }
        __syncthreads();
        double *out = a.slab + (int64_t)blockIdx.x * a.ld;
        double rs = 0.0;
        // (the observations' slots in batches of eight: one round trip per batch, not per observation)
        for (int64_t i0 = 0; i0 < a.ld; i0 += 8 * T) {
            int off[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int64_t i = i0 + q * T + tid;
                off[q] = i < g.N ? g.lds_of[i] : -1;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int64_t i = i0 + q * T + tid;
                if (i < a.ld) {
                    const double v = off[q] >= 0 ? Tc[off[q]] : 0.0;
                    out[i] = v;
                    rs += v;
                }
            }
        }
        if (a.dsum) {
            // (sum of this workgroup's slab row: the epilogue then knows mean(d) up front and needs one launch)
            const double t = block_allreduce_sum(rs, red, NWV);
            if (tid == 0) a.dsum[blockIdx.x] = t;
        }
    }
    mark(5);
    if (clk)
        for (int q = 0; q < 8; ++q) g.dbg[q] += tph_s[q];
}

// wm_j = (sum_i K_ij^2)^wf for j = (c, k): thread per cell, observations in order (serial, fixed order)
__global__ void __launch_bounds__(256)
lonsym_colnorm_kernel(LonSymGeom g, const int *__restrict__ a_of, const int *__restrict__ m_of, double wf,
                      double *__restrict__ wm)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= (int64_t)g.nc * g.n) return;
    const int c = (int)(j / g.n), k = (int)(j - (int64_t)c * g.n);
    const double *Tg = g.T + (int64_t)c * g.ldT;
    double s = 0.0;
    for (int64_t i = 0; i < g.N; ++i) {
        int d = m_of[i] - k;
        if (d < 0) d += g.n;
        const double v = Tg[a_of[i] * g.n + d];
        s += v * v;
    }
    wm[j] = (wf == 0.5) ? sqrt(s) : pow(s, wf);
}

}  // namespace ghk
