// Several HMC chains sharing every sweep of G: the fp64 MFMA path (gfx950, wave64).
//
// With C chains (C <= 16) the two products of a potential evaluation are skinny GEMMs,
//     S (M x 16) = G^T (M x N) . R (N x 16)        adjoint of all chains
//     D (N x 16) = G   (N x M) . X (M x 16)        forward of all chains
// i.e. 2*16 flop per byte of G instead of 0.5: v_mfma_f64_16x16x4_f64 does the contraction and
// the cross-lane sums, G is still read once per product.  (The single-chain path keeps one read
// of G per leapfrog step by holding a column between the dot product and the axpy; with 16
// chains a 16-column tile of G is 1.3 MB at C2 and cannot stay on chip, so the batch uses two
// sweeps per step -- 1/8 of a sweep per chain-step.)
//
// Layouts: chain-interleaved vectors X[j][c], P[j][c], GREG[j][c] (M x 16) and D[i][c] (N x 16);
// residuals patch-transposed Rt[i/16][h][k][c][t] with row i%16 = 8 h + 2 k + t, so that the lane
// that feeds row-group k of the MFMAs reads its rows {2k, 2k+1} and {8+2k, 8+2k+1} as two 16-byte
// words and the four lanes of a column cover one contiguous 64-byte half line per load.
//
// MFMA operand maps (cdna guide, f64 16x16x4): A lane l -> A[l&15][l>>4], B lane l -> B[l>>4][l&15],
// C/D lane l, reg q -> D[(l>>4) + 4q][l&15].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ghk {

using d4 = double __attribute__((ext_vector_type(4)));
constexpr int CB = 16;  // chain slots of a batch (MFMA N dimension)

// PH_PFIN_SPEC: the final half momentum step of a trajectory AND, from the same gradient at the
// proposal, the first leapfrog step of the chain's next trajectory with its momentum Pn (valid if
// the proposal is accepted; a rejected one simply discards it): a trajectory then costs L sweep
// pairs instead of L + 1.
enum : int { PH_IDLE = 0, PH_UPD = 1, PH_PFIN = 2, PH_GOUT = 3, PH_PFIN_SPEC = 4 };

struct BatchAdjArgs {
    const double *Gb;       // MFMA-operand-ordered copy of G (see batch_relayout_kernel) or nullptr
    const double *G;
    int64_t ld, M;
    int np;                 // row patches of 16: ld / 16
    const double *Rt;       // np x 2 x 4 x 16 x 2
    const double *GREG;     // M x 16 (alpha * grad R per chain)
    const double *X_in, *P_in;
    const double *Pn;       // M x 16 momenta of the next trajectories (PH_PFIN_SPEC) or nullptr
    double *X_out, *P_out;  // M x 16
    const double *low, *high;  // M
    double *G_out;          // M x 16 gradient (PH_GOUT) or nullptr
    double *pp_part;        // n_waves x 16: sum of p^2 after the final half step (PH_PFIN)
    int phase[CB];
    double cu[CB], cp[CB];
    double dt;
    int n_waves;            // total waves of the launch
};

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Adjoint of all chains + leapfrog update (hmc.py:114-152): every wave owns TWO adjacent
// 16-column tiles, so each fragment of the residuals read from L2 feeds two MFMAs.
// Rows 16 p + 2 k + {0,1} and 16 p + 8 + 2 k + {0,1} of patch p are contracted by the four MFMAs
// of lane group k = lane >> 4.
__global__ void __launch_bounds__(256) batch_adjoint_kernel(BatchAdjArgs a)
{
    const int lane = threadIdx.x & 63;
    const int lo = lane & 15, k = lane >> 4;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t ntiles = (a.M + 15) / 16;
    const int64_t npairs = (ntiles + 1) / 2;
    const bool tiled = a.Gb != nullptr;
    double pp = 0.0;
    for (int64_t pair = wave; pair < npairs; pair += a.n_waves) {
        // column-major G: 16 columns x 64 B per load.  Gb: the tile's operands are one contiguous
        // stream (1 KiB per load), which is what keeps HBM pages open.
        const d2 *gcol[2];
        bool tile_ok[2], col_ok[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t tile = 2 * pair + h;
            const int64_t ja = tile * 16 + lo;  // the column this lane feeds as A operand
            tile_ok[h] = tile < ntiles;
            col_ok[h] = tile_ok[h] && (tiled || ja < a.M);
            gcol[h] = tiled ? reinterpret_cast<const d2 *>(a.Gb) + (tile_ok[h] ? tile : 0) * a.np * 128 + lane
                            : reinterpret_cast<const d2 *>(a.G + (col_ok[h] ? ja : 0) * a.ld) + k;
        }
        const int gstep = tiled ? 128 : 8, ghalf = tiled ? 64 : 4;
        const d2 *rt = reinterpret_cast<const d2 *>(a.Rt) + (k * 16 + lo);
        d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
        // software pipeline: two patches are in flight while one is multiplied (ring of three)
        d2 g0[3][2], g1[3][2], r0[3], r1[3];
        // (every load unconditional -- patches past the end re-read the last one, columns past M the
        // first, and the residual fragment is zeroed AFTER it arrived: one conditional load here and
        // the compiler waits with vmcnt(0) in front of every multiply, so nothing is in flight
        // while a patch is multiplied; see sweep_kernel)
        const int plast = a.np - 1;
        auto fetch = [&](int slot, int p) {
            const bool ok = p < a.np;
            const int pc = ok ? p : plast;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                g0[slot][h] = __builtin_nontemporal_load(gcol[h] + (int64_t)gstep * pc);
                g1[slot][h] = __builtin_nontemporal_load(gcol[h] + (int64_t)gstep * pc + ghalf);
            }
            const d2 ra = rt[128 * pc], rb = rt[128 * pc + 64];
            r0[slot] = ok ? ra : d2{0.0, 0.0};
            r1[slot] = ok ? rb : d2{0.0, 0.0};
        };
        auto mult = [&](int slot) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                acc[h] = mfma_f64(g0[slot][h].x, r0[slot].x, acc[h]);
                acc[h] = mfma_f64(g0[slot][h].y, r0[slot].y, acc[h]);
                acc[h] = mfma_f64(g1[slot][h].x, r1[slot].x, acc[h]);
                acc[h] = mfma_f64(g1[slot][h].y, r1[slot].y, acc[h]);
            }
        };
        fetch(0, 0);
        fetch(1, 1);
        for (int p = 0; p < a.np; p += 3) {  // zero patches beyond np add nothing
            fetch(2, p + 2);
            mult(0);
            fetch(0, p + 3);
            mult(1);
            fetch(1, p + 4);
            mult(2);
        }
        // acc[h][q] = <G_j, r_c> for column j = 16 (2 pair + h) + k + 4 q and chain c = lo
        const int c = lo;
        const int ph = a.phase[c];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (!tile_ok[h]) continue;
            const int64_t j0 = (2 * pair + h) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t j = j0 + k + 4 * q;
                if (j >= a.M) continue;
                const int64_t idx = j * CB + c;
                const double g = 2.0 * acc[h][q] + (a.GREG ? a.GREG[idx] : 0.0);
                if (ph == PH_GOUT) {
                    a.G_out[idx] = g;
                } else if (ph == PH_UPD) {
                    double pj = a.P_in[idx] - a.cu[c] * g;
                    double xj = a.X_in[idx] + a.dt * pj;
                    const double hi = a.high[j], lw = a.low[j];
                    if (xj > hi) {
                        xj = hi;
                        pj = -pj;
                    } else if (xj < lw) {
                        xj = lw;
                        pj = -pj;
                    }
                    a.P_out[idx] = pj;
                    a.X_out[idx] = xj;
                } else if (ph == PH_PFIN) {
                    const double pf = a.P_in[idx] - a.cp[c] * g;
                    pp += pf * pf;
                    a.P_out[idx] = pf;
                    a.X_out[idx] = a.X_in[idx];
                } else if (ph == PH_PFIN_SPEC) {
                    const double pf = a.P_in[idx] - a.cp[c] * g;
                    pp += pf * pf;
                    double pj = a.Pn[idx] - a.cu[c] * g;
                    double xj = a.X_in[idx] + a.dt * pj;
                    const double hi = a.high[j], lw = a.low[j];
                    if (xj > hi) {
                        xj = hi;
                        pj = -pj;
                    } else if (xj < lw) {
                        xj = lw;
                        pj = -pj;
                    }
                    a.P_out[idx] = pj;
                    a.X_out[idx] = xj;
                } else {
                    a.P_out[idx] = a.P_in[idx];
                    a.X_out[idx] = a.X_in[idx];
                }
            }
        }
    }
    // lanes lo, lo+16, lo+32, lo+48 hold parts of chain lo: combine in fixed order
    pp += __shfl_xor(pp, 16, WAVE);
    pp += __shfl_xor(pp, 32, WAVE);
    // only chains that took their final half step in THIS sweep write: the slot keeps that value
    if (a.pp_part && lane < 16 && wave < a.n_waves && (a.phase[lane] == PH_PFIN || a.phase[lane] == PH_PFIN_SPEC))
        a.pp_part[(int64_t)wave * CB + lane] = pp;
}

// Gb[tile][patch][h][lane] (16-byte words): rows 16 p + 8 h + 2 k + {0,1} of column 16 tile + lo,
// lane = lo + 16 k -- exactly what lane `lane` feeds to the MFMAs of batch_adjoint_kernel, so a
// wave streams its tile as one contiguous run.  Zero beyond M.
__global__ void __launch_bounds__(256)
batch_relayout_kernel(const double *G, int64_t ld, int64_t M, int np, int64_t ntiles, double *Gb)
{
    const int64_t total = ntiles * np * 128;  // 16-byte words
    d2 *out = reinterpret_cast<d2 *>(Gb);
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int lane = (int)(t & 63), h = (int)((t >> 6) & 1);
        const int64_t tp = t >> 7;
        const int64_t tile = tp / np;
        const int p = (int)(tp - tile * np);
        const int64_t col = tile * 16 + (lane & 15);
        const int64_t row = 16 * p + 8 * h + 2 * (lane >> 4);
        out[t] = (col < M) ? *reinterpret_cast<const d2 *>(G + col * ld + row) : d2{0.0, 0.0};
    }
}

struct BatchFwdArgs {
    const double *G;
    int64_t ld, M, N;
    const double *X;         // M x 16
    int64_t cols_per_block;  // multiple of 16
    double *slab;            // gridDim.y x (ld x 16)
};

// Forward of all chains: workgroup = 4 waves = 512 rows, blockIdx.y = block of columns.  Each wave
// keeps four 32-row patches (two MFMA tiles each: rows 2i and 2i+1 of a lane's 16-byte load).
__global__ void __launch_bounds__(256) batch_forward_kernel(BatchFwdArgs a)
{
    const int lane = threadIdx.x & 63;
    const int lo = lane & 15, k = lane >> 4;
    const int w = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * 512 + w * 128;  // first row of this wave
    const int64_t jb0 = (int64_t)blockIdx.y * a.cols_per_block;
    int64_t jb1 = jb0 + a.cols_per_block;
    if (jb1 > a.M) jb1 = a.M;
    d4 acc[4][2];
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
        acc[rp][0] = d4{0.0, 0.0, 0.0, 0.0};
        acc[rp][1] = d4{0.0, 0.0, 0.0, 0.0};
    }
    // rows this lane loads in patch rp: i0 + 32 rp + 2 lo, +1
    // pipeline unit = two groups of 4 columns (8 loads of 16 B, 16 MFMAs); the next unit is in
    // flight while the current one is multiplied
    double bA[2], bB[2];
    d2 gA[2][4], gB[2][4];
    // (every load unconditional: columns past the block re-read its last column with a zero
    // multiplier, rows past the end the column's last double2 -- their sums are never stored; a
    // conditional load would make every wait in front of the MFMAs a vmcnt(0), see sweep_kernel)
    const int64_t jlast = (jb1 > jb0 ? jb1 : a.M) - 1;
    int64_t roff[4];
#pragma unroll
    for (int rp = 0; rp < 4; ++rp) {
        const int64_t r = i0 + 32 * rp + 2 * lo;
        roff[rp] = (r + 1 < a.ld) ? r : a.ld - 2;
    }
    auto fetch = [&](double (&b)[2], d2 (&g)[2][4], int64_t jbase) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t j = jbase + 4 * u + k;
            const bool ok = j < jb1;
            const int64_t jc = ok ? j : jlast;
            const double xv = a.X[jc * CB + lo];
            b[u] = ok ? xv : 0.0;
            const double *col = a.G + jc * a.ld;
#pragma unroll
            for (int rp = 0; rp < 4; ++rp)
                g[u][rp] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(col + roff[rp]));
        }
    };
    auto mult = [&](const double (&b)[2], const d2 (&g)[2][4]) {
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int rp = 0; rp < 4; ++rp) {
                acc[rp][0] = mfma_f64(g[u][rp].x, b[u], acc[rp][0]);
                acc[rp][1] = mfma_f64(g[u][rp].y, b[u], acc[rp][1]);
            }
    };
    fetch(bA, gA, jb0);
    for (int64_t j0 = jb0; j0 < jb1; j0 += 16) {
        fetch(bB, gB, j0 + 8);
        mult(bA, gA);
        fetch(bA, gA, j0 + 16);
        mult(bB, gB);
    }
    // acc[rp][s][q]: row i0 + 32 rp + 2 (k + 4 q) + s, chain lo
    double *out = a.slab + (int64_t)blockIdx.y * a.ld * CB;
#pragma unroll
    for (int rp = 0; rp < 4; ++rp)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row = i0 + 32 * rp + 2 * (k + 4 * q) + s;
                if (row < a.ld) out[row * CB + lo] = acc[rp][s][q];
            }
}

// D[i][c] = sum over column blocks (fixed order)
__global__ void __launch_bounds__(256)
batch_reduce_kernel(const double *slab, int nblocks, int64_t n, double *D)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n) return;
    double s = 0.0;
    for (int b = 0; b < nblocks; ++b) s += slab[(int64_t)b * n + t];
    D[t] = s;
}

struct BatchRegArgs {
    int kind;
    int64_t M;
    int nz, ny, nx;
    double alpha, beta;
    const double *X, *mwapr, *wm2;
    double *GREG;     // M x 16
    double *regpart;  // gridDim.x x 16
};

// Regulariser of every chain: block = 16 cells x 16 chains.
__global__ void __launch_bounds__(256) batch_reg_kernel(BatchRegArgs a)
{
    __shared__ double red[16][17];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int64_t j = (int64_t)blockIdx.x * 16 + q;
    double val = 0.0;
    if (j < a.M) {
        const double v = a.X[j * CB + c] - a.mwapr[j];
        double g = 0.0;
        if (a.kind == 0) {
            val = v * v;
            g = 2.0 * v;
        } else if (a.kind == 2) {
            const double v2 = v * v, den = v2 + a.beta, w2 = a.wm2[j];
            val = (w2 * v2) / den;
            g = (2.0 * a.beta * w2 * v) / (den * den);
        } else {
            const int64_t nx = a.nx, ny = a.ny, nz = a.nz;
            const int64_t i = j % nx, jj = (j / nx) % ny, kk = j / (nx * ny);
            const int64_t stride[3] = {1, nx, nx * ny};
            const bool fwd[3] = {i < nx - 1, jj < ny - 1, kk < nz - 1};
            const bool bwd[3] = {i > 0, jj > 0, kk > 0};
#pragma unroll
            for (int ax = 0; ax < 3; ++ax) {
                if (fwd[ax]) {
                    const int64_t n = j + stride[ax];
                    const double t = v - (a.X[n * CB + c] - a.mwapr[n]);
                    if (a.kind == 1) {
                        val += t * t;
                        g += 2.0 * t;
                    } else {
                        const double s = sqrt(t * t + a.beta);
                        val += s;
                        g += t / s;
                    }
                }
                if (bwd[ax]) {
                    const int64_t n = j - stride[ax];
                    const double t = (a.X[n * CB + c] - a.mwapr[n]) - v;
                    if (a.kind == 1)
                        g -= 2.0 * t;
                    else
                        g -= t / sqrt(t * t + a.beta);
                }
            }
        }
        a.GREG[j * CB + c] = a.alpha * g;
    }
    red[q][c] = val;
    __syncthreads();
    if (q == 0) {
        double s = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) s += red[r][c];
        a.regpart[(int64_t)blockIdx.x * CB + c] = s;
    }
}

struct BatchFinishArgs {
    int64_t N, ld;
    int n_regpart;
    const double *D;        // ld x 16
    const double *gfix, *dobs_c;
    const double *regpart;  // n_regpart x 16
    double alpha;
    double *Rt;             // patch-transposed residuals
    double *scal;           // 16 x 4: U_data, R, U, mean
};

// One workgroup per chain: mean removal, residual, data misfit, U (potential.py:700-706).
__global__ void __launch_bounds__(1024) batch_finish_kernel(BatchFinishArgs a)
{
    __shared__ double red[16];
    const int c = blockIdx.x;
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < a.N; i += 1024) s += a.D[i * CB + c] + (a.gfix ? a.gfix[i] : 0.0);
    const double mean = block_allreduce_sum(s, red, 16) / (double)a.N;
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < a.ld; i += 1024) {
        double ri = 0.0;
        if (i < a.N) {
            const double dinv = a.D[i * CB + c] + (a.gfix ? a.gfix[i] : 0.0);
            ri = (dinv - mean) - a.dobs_c[i];
            acc += ri * ri;
        }
        const int64_t p = i >> 4, q = i & 15;  // row q = 8 h + 2 k + t
        a.Rt[(((p * 2 + (q >> 3)) * 4 + ((q & 7) >> 1)) * CB + c) * 2 + (q & 1)] = ri;
    }
    const double ud = block_allreduce_sum(acc, red, 16);
    double rs = 0.0;
    for (int t = threadIdx.x; t < a.n_regpart; t += 1024) rs += a.regpart[(int64_t)t * CB + c];
    const double R = block_allreduce_sum(rs, red, 16);
    if (threadIdx.x == 0) {
        a.scal[c * 4 + 0] = ud;
        a.scal[c * 4 + 1] = R;
        a.scal[c * 4 + 2] = ud + a.alpha * R;
        a.scal[c * 4 + 3] = mean;
    }
}

// per-block, per-chain partial sums of P[j][c]^2 (block = 16 cells x 16 chains, grid-stride)
__global__ void __launch_bounds__(256)
batch_sumsq_kernel(const double *P, int64_t M, double *part)
{
    __shared__ double red[16][17];
    const int c = threadIdx.x & 15, q = threadIdx.x >> 4;
    double s = 0.0;
    for (int64_t j = (int64_t)blockIdx.x * 16 + q; j < M; j += (int64_t)gridDim.x * 16) {
        const double v = P[j * CB + c];
        s += v * v;
    }
    red[q][c] = s;
    __syncthreads();
    if (q == 0) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += red[r][c];
        part[(int64_t)blockIdx.x * CB + c] = t;
    }
}

// chain-major host layout (C x M) <-> chain-interleaved device layout (M x 16)
__global__ void __launch_bounds__(256)
batch_interleave_kernel(const double *rows, int C, int64_t M, double *out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * CB) return;
    const int c = (int)(t & 15);
    const int64_t j = t >> 4;
    out[t] = (c < C) ? rows[(int64_t)c * M + j] : 0.0;
}

__global__ void __launch_bounds__(256)
batch_extract_kernel(const double *X, int c, int64_t M, double *out)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j < M) out[j] = X[j * CB + c];
}

// rows (chain-major, C x M) of the chains in `mask` into their columns of an interleaved array
__global__ void __launch_bounds__(256)
batch_scatter_kernel(const double *rows, int64_t M, unsigned mask, double *out)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= M * CB) return;
    const int c = (int)(t & 15);
    if (mask & (1u << c)) out[t] = rows[(int64_t)c * M + (t >> 4)];
}

// copy the columns of the accepted chains from the proposal into the current state
__global__ void __launch_bounds__(256)
batch_commit_kernel(const double *src, double *dst, int64_t n16, unsigned mask)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n16) return;
    if (mask & (1u << (t & 15))) dst[t] = src[t];
}

// the same for the patch-transposed residuals (chain index sits in bits 1..4 of the offset)
__global__ void __launch_bounds__(256)
batch_commit_rt_kernel(const double *src, double *dst, int64_t n16, unsigned mask)
{
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n16) return;
    if (mask & (1u << ((t >> 1) & 15))) dst[t] = src[t];
}

}  // namespace ghk
