// Several HMC chains on the matrix-free kernel: every evaluated entry serves all chains (gfx950).
//
// BASELINE configs[3] is "matrix-free, 8 chains per node x 8 GPUs" (example/global/run_main.sh:16,
// inversion/hmc.py:367-369: the reference runs its chains as separate MPI ranks, each re-evaluating
// the whole tesseroid kernel, gravmag/_tesseroid_numba.py:32-71).  The single-chain matrix-free
// pass (mf_tess_fast_kernel / mf_fused_kernel) is bound by evaluating the entries (~127 VALU
// instructions each), not by using them (2 FMAs).  With C <= 16 chains in lock step the two
// products of a potential evaluation are skinny GEMMs whose G operand is computed on the fly,
//     S (M x 16) = K^T (M x N) . R (N x 16)       adjoint of all chains + leapfrog update
//     D (N x 16) = K   (N x M) . XS (M x 16)      forward of all chains, XS = X / wm
// and v_mfma_f64_16x16x4 does the per-chain work: a pass costs the same for 1 or 16 chains.
//
// The one-evaluation fusion of the single-chain pass (a workgroup keeps a column between the dot
// and the axpy) does not carry over: between the dot and the axpy sits the update, which needs the
// dot over ALL rows, and the per-workgroup state (C x N residuals + C x N forward partials) no
// longer fits one CU.  So the batch evaluates every entry twice per step -- for ALL chains.
//
// Both passes share the evaluation phase: a workgroup of 16 waves stages a tile of 16 columns x 512
// rows in LDS, wave w evaluating column w of the tile with the lane = row layout and the column's
// constants in scalar registers (the single-chain pass's arithmetic: tess_leaf_fast / tess_leaf_cc
// / tess_entry_cc / prism_entry, unchanged).  Two staging buffers: the next chunk is evaluated
// while the MFMAs consume this one, one barrier per chunk.  The rows' observation constants come
// from LDS as well (fetched by the whole workgroup one chunk ahead): as per-lane global loads in
// front of every evaluation -- 16 waves x the same rows -- the compiler's software pipeline ended up
// waiting for half of them right after issuing them (VALU busy 72 %, profiles/r03).
//   adjoint: workgroup = column tile, all rows in chunks; lane (lo, k) feeds A[col lo][row group k]
//            from the staged tile (16-byte LDS reads), B from the patch-transposed residuals Rt of
//            batch.hip.h; the 16 waves' accumulators are summed through LDS in wave order, then
//            256 threads apply the leapfrog update of the 16 x 16 (cell, chain) pairs.
//   forward: workgroup = 512-row chunk x range of column tiles; wave w owns row patches 2w, 2w+1
//            of the chunk (two accumulator tiles), A[row lo][col k] from the staged tile, B = XS.
// Every sum has a fixed order: results are reproducible bit for bit.
//
// Tesseroids with the near-field table (KIND 2, 3): the staged value of EVERY pair is its root
// leaf; the listed pairs contribute their difference delta = entry - root leaf through two small
// sparse kernels (column-major list for the adjoint, row-major copy for the forward), so the
// dense passes have no test, no branch and no scattered LDS writes.
#pragma once

namespace ghk {

constexpr int MFB_WAVES = 16;             // waves per workgroup = columns of a tile
// Row blocks of 64 per staged chunk.  Forward: 8 (512 rows; the workgroup's rows never change: their
// observation constants sit in LDS once).  Adjoint: 7 (448 rows), which leaves room for two buffers of
// observation constants: the next chunk's are fetched while this one is evaluated.
constexpr int MFB_RC_FWD = 8, MFB_RC_ADJ = 7;
constexpr int MFB_NOBS = 5;               // observation constants per row (tess_leaf_fast: 5)
// column stride of the staged tile (doubles): +8 shifts consecutive columns by 64 B, so the
// adjoint's 16-byte reads (16 columns x 4 row groups) and the forward's 8-byte reads (4 columns x
// 16 rows) both spread over all LDS banks
template <int RC> struct MfbTile {
    static constexpr int ROWS = RC * 64;
    static constexpr int PATCHES = ROWS / 16;
    static constexpr int S = ROWS + 8;
    static constexpr int BUF = MFB_WAVES * S;   // doubles per staging buffer
};
constexpr size_t MFB_LDS_FWD = (2 * (size_t)MfbTile<MFB_RC_FWD>::BUF + (size_t)MFB_NOBS * MfbTile<MFB_RC_FWD>::ROWS) * sizeof(double);      // 153600 B
constexpr size_t MFB_LDS_ADJ = (2 * (size_t)MfbTile<MFB_RC_ADJ>::BUF + 2 * (size_t)MFB_NOBS * MfbTile<MFB_RC_ADJ>::ROWS) * sizeof(double);  // 152576 B

// the column's constants in (scalar) registers
// KIND: 0 prisms; 1 tesseroids with the subdivision inside the pass; 2 / 3 tesseroids with the near-field
// list and the reference-order / the fast root leaf; 4 = 3 with every observation at one height: what
// depends on (radius, cell) alone is formed once per column (10 of an entry's 97 instructions).
template <int KIND>
struct MfbCol {
    double cc[KIND == 0 ? 1 : TESS_NC];
    double b[KIND == 0 ? 6 : 1];
    double pre[KIND == 4 ? 8 : 1];
    const double *ccp, *bp;  // KIND 1: the generic engine takes pointers
};

template <int KIND>
__device__ __forceinline__ void mfb_col_load(MfbCol<KIND> &c, const MfGeom &g, const double *cellc, int64_t j)
{
    if constexpr (KIND == 0) {
        const kconst_ptr bk = as_kconst(g.bounds6) + 6 * j;
#pragma unroll
        for (int q = 0; q < 6; ++q) c.b[q] = bk[q];
        c.ccp = c.bp = nullptr;
    } else {
        const kconst_ptr ck = as_kconst(cellc) + (int64_t)TESS_NC * j;
        constexpr int Q0 = KIND >= 3 ? 10 : (KIND == 2 ? 8 : 0);
        constexpr int Q1 = KIND >= 3 ? TESS_NC : 23;
#pragma unroll
        for (int q = Q0; q < Q1; ++q) c.cc[q] = ck[q];
        c.ccp = cellc + (int64_t)TESS_NC * j;
        c.bp = g.bounds6 + 6 * j;
        if constexpr (KIND == 4) tess_leaf_fast_pre(g.radius_u, c.cc, c.pre);
    }
}

// observation constants of a row: KIND 0: x, y, z; 1, 2: lon, sin lat, cos lat, radius; 3: sin lon,
// cos lon, sin lat, cos lat, radius.  The q-th array they come from:
template <int KIND>
__device__ __forceinline__ const double *mfb_obs_array(const MfGeom &g, int q)
{
    if constexpr (KIND >= 3) return q == 0 ? g.o4 : q == 1 ? g.o5 : q == 2 ? g.o1 : q == 3 ? g.o2 : g.o3;
    return q == 0 ? g.o0 : q == 1 ? g.o1 : q == 2 ? g.o2 : g.o3;
}
template <int KIND> constexpr int mfb_nobs() { return KIND == 0 ? 3 : KIND == 3 ? 5 : 4; }  // (KIND 4: no radius)

// The constants of `rows` rows starting at row0, fetched by the whole workgroup (rows past the end
// re-read the last row: finite values whose products meet zero residuals / are never stored): each
// thread requests its elements (fetch) and parks them in LDS later (park), SoA: ob[q * rows + i].
template <int KIND, int ROWS>
struct MfbObsFetch {
    static constexpr int PER = (mfb_nobs<KIND>() * ROWS + 1023) / 1024;
    double v[PER];
    __device__ __forceinline__ void fetch(const MfGeom &g, int64_t row0, int tid)
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + k * 1024;
            const int q = e / ROWS, i = e - q * ROWS;
            int64_t row = row0 + i;
            if (row >= g.N) row = g.N - 1;
            v[k] = q < mfb_nobs<KIND>() ? mfb_obs_array<KIND>(g, q)[row] : 0.0;
        }
    }
    __device__ __forceinline__ void park(double *ob, int tid) const
    {
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int e = tid + k * 1024;
            if (e < mfb_nobs<KIND>() * ROWS) ob[e] = v[k];
        }
    }
};

template <int KIND>
__device__ __forceinline__ double mfb_eval(const MfbCol<KIND> &c, const MfGeom &g, const double (&o)[5], unsigned &nleaf)
{
    if constexpr (KIND == 0) {
        nleaf += 1;
        return prism_entry(o[0], o[1], o[2], c.b);
    } else if constexpr (KIND == 1) {
        return tess_entry_cc(o[0], o[1], o[2], o[3], c.ccp, c.bp, g.ratio, nleaf);
    } else if constexpr (KIND == 2) {
        nleaf += 1;
        return tess_leaf_cc(o[0], o[1], o[2], o[3], c.cc);
    } else if constexpr (KIND == 3) {
        nleaf += 1;
        return tess_leaf_fast(o[0], o[1], o[2], o[3], o[4], c.cc);
    } else {
        nleaf += 1;
        return tess_leaf_fast_ru(o[0], o[1], o[2], o[3], c.cc, c.pre);
    }
}

// One wave stages its column for `nb` row blocks of a chunk: st[e * 64 + lane]; the rows' constants
// come from the chunk's LDS copy ob[q * ROWS + e * 64 + lane].  between(e) runs after evaluation e:
// the MFMAs of the PREVIOUS chunk / tile ride there, one per evaluation -- issued to the matrix
// pipe, they execute in the shadow of the next evaluation's VALU work.  (All of them right behind the
// barrier, every wave at once, left the VALU idle for 13 % of the pass: fp64 MFMA takes 16 passes on
// gfx950.)
template <int KIND, int ROWS, typename F>
__device__ __forceinline__ void mfb_stage(const MfbCol<KIND> &col, const MfGeom &g, const double *ob, int nb, int lane,
                                          double *st, unsigned &nleaf, F between)
{
    if constexpr (KIND == 0 || KIND == 1) {
        // prisms (8 corners x (sqrt, 2 log, atan2)) and the adaptive tesseroid engine: hundreds of
        // instructions per entry -- one copy of the evaluation, rolled (fifteen unrolled copies spilled 200 ..
        // 550 registers); the selects of between(e) by a run-time e are noise here
#pragma unroll 1
        for (int e = 0; e < nb; ++e) {
            double o[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) o[q] = q < mfb_nobs<KIND>() ? ob[q * ROWS + e * 64 + lane] : 0.0;
            st[e * 64 + lane] = mfb_eval<KIND>(col, g, o, nleaf);
            between(e);
        }
        return;
    }
    // (unrolled: between(e) picks its operands by e -- as a run-time index that is a chain of selects;
    // a full chunk, the usual case, without a branch per evaluation: the next evaluation's LDS reads
    // may then move up into this one instead of being waited for at its start)
    if (nb == ROWS / 64) {
        double o[2][5];
#pragma unroll
        for (int q = 0; q < 5; ++q) o[0][q] = q < mfb_nobs<KIND>() ? ob[q * ROWS + lane] : 0.0;
#pragma unroll
        for (int e = 0; e < ROWS / 64; ++e) {
            if (e + 1 < ROWS / 64) {
#pragma unroll
                for (int q = 0; q < 5; ++q) o[(e + 1) & 1][q] = q < mfb_nobs<KIND>() ? ob[q * ROWS + (e + 1) * 64 + lane] : 0.0;
            }
            __builtin_amdgcn_sched_barrier(0);  // (the reads above stay in front of this evaluation)
            st[e * 64 + lane] = mfb_eval<KIND>(col, g, o[e & 1], nleaf);
            between(e);
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < ROWS / 64; ++e) {
        if (e < nb) {
            double o[5];
#pragma unroll
            for (int q = 0; q < 5; ++q) o[q] = q < mfb_nobs<KIND>() ? ob[q * ROWS + e * 64 + lane] : 0.0;
            st[e * 64 + lane] = mfb_eval<KIND>(col, g, o, nleaf);
            between(e);
        }
    }
}

__device__ __forceinline__ void mfb_count(MfStats *stats, unsigned nent, unsigned nleaf, int lane)
{
    if (!stats) return;
    unsigned long long e = nent, l = nleaf;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        e += __shfl_xor(e, off, WAVE);
        l += __shfl_xor(l, off, WAVE);
    }
    if (lane == 0) {
        atomicAdd(&stats->entries, e);
        atomicAdd(&stats->leaves, l);
    }
}

// Leapfrog update of one (cell, chain) pair from its gradient gr, by the chain's phase (the epilogue of
// batch_adjoint_kernel; hmc.py:114-152).  write: this thread stores the results and counts p^2.
// Returns the position the forward product of the next evaluation uses.
__device__ __forceinline__ double mfb_update(const BatchAdjArgs &a, int64_t idx, int64_t jc, int c, double gr, bool write,
                                             double &pp)
{
    const int ph = a.phase[c];
    double xo = a.X_in[idx], po = a.P_in[idx];
    if (ph == PH_GOUT) {
        if (write) a.G_out[idx] = gr;
        return xo;
    }
    if (ph == PH_UPD || ph == PH_PFIN_SPEC) {
        if (ph == PH_PFIN_SPEC) {
            const double pf = po - a.cp[c] * gr;
            if (write) pp += pf * pf;
            po = a.Pn[idx];
        }
        double pj = po - a.cu[c] * gr;
        double xj = xo + a.dt * pj;
        const double hi = a.high[jc], lw = a.low[jc];
        if (xj > hi) {
            xj = hi;
            pj = -pj;
        } else if (xj < lw) {
            xj = lw;
            pj = -pj;
        }
        po = pj;
        xo = xj;
    } else if (ph == PH_PFIN) {
        const double pf = po - a.cp[c] * gr;
        if (write) pp += pf * pf;
        po = pf;
    }
    if (write) {
        a.P_out[idx] = po;
        a.X_out[idx] = xo;
    }
    return xo;
}

// The same with the pair's operands requested ahead (the fused team pass: the loads are in flight while
// the adjoint MFMAs run and the team's parts are collected).
struct MfbUpd {
    double x, p, greg, pn, hi, lo, snear, iw;
};

__device__ __forceinline__ void mfb_upd_load(MfbUpd &u, const BatchAdjArgs &a, const double *Snear, const double *iw,
                                             int64_t idx, int64_t jc)
{
    u.x = a.X_in[idx];
    u.p = a.P_in[idx];
    u.greg = a.GREG ? a.GREG[idx] : 0.0;
    u.pn = a.Pn ? a.Pn[idx] : 0.0;
    u.hi = a.high[jc];
    u.lo = a.low[jc];
    u.snear = Snear ? Snear[idx] : 0.0;
    u.iw = iw[jc];
}

// (ph, cu, cp: the chain's phase and coefficients, read once per launch -- indexed by the thread they
// are loads from the kernel-argument segment, a memory round trip in front of every tile's update)
__device__ __forceinline__ double mfb_update_pre(const BatchAdjArgs &a, const MfbUpd &u, int64_t idx, int ph, double cu,
                                                 double cp, double gr, bool write, double &pp)
{
    double xo = u.x, po = u.p;
    if (ph == PH_GOUT) {
        if (write) a.G_out[idx] = gr;
        return xo;
    }
    if (ph == PH_UPD || ph == PH_PFIN_SPEC) {
        if (ph == PH_PFIN_SPEC) {
            const double pf = po - cp * gr;
            if (write) pp += pf * pf;
            po = u.pn;
        }
        double pj = po - cu * gr;
        double xj = xo + a.dt * pj;
        if (xj > u.hi) {
            xj = u.hi;
            pj = -pj;
        } else if (xj < u.lo) {
            xj = u.lo;
            pj = -pj;
        }
        po = pj;
        xo = xj;
    } else if (ph == PH_PFIN) {
        const double pf = po - cp * gr;
        if (write) pp += pf * pf;
        po = pf;
    }
    if (write) {
        a.P_out[idx] = po;
        a.X_out[idx] = xo;
    }
    return xo;
}

// ---- adjoint of all chains + leapfrog update (hmc.py:114-152) ------------------------------------
// BatchAdjArgs as batch_adjoint_kernel (G / Gb unused); iw = 1 / wm (1 where wm == 0), Snear =
// near-field part of S (M x 16) or nullptr; pp_part has gridDim.x x 16 entries.
template <int KIND>
__global__ void __launch_bounds__(1024)
mfb_adjoint_kernel(MfGeom g, BatchAdjArgs a, const double *__restrict__ iw, const double *__restrict__ cellc,
                   const double *__restrict__ Snear, MfStats *stats)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double ppred[16][17];
    using TL = MfbTile<MFB_RC_ADJ>;
    double *obs_s = smem + 2 * TL::BUF;  // 2 x (MFB_NOBS x ROWS): constants of this / the next chunk's rows
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, k = lane >> 4;
    const int64_t ntiles = (a.M + 15) / 16;
    const int nrb = (int)((a.ld + 63) / 64);
    const int nch = (nrb + MFB_RC_ADJ - 1) / MFB_RC_ADJ;
    const d2 *rt = reinterpret_cast<const d2 *>(a.Rt) + (k * 16 + lo);
    double pp = 0.0;
    unsigned nent = 0, nleaf = 0;
    MfbObsFetch<KIND, TL::ROWS> of;
    of.fetch(g, 0, tid);
    of.park(obs_s, tid);
    int oi = 0;  // obs buffer of the chunk being staged
    __syncthreads();
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        int64_t j = tile * 16 + wave;
        if (j >= a.M) j = a.M - 1;
        MfbCol<KIND> col;
        mfb_col_load<KIND>(col, g, cellc, j);
        d4 acc = d4{0.0, 0.0, 0.0, 0.0};
        // MFMAs of the chunk staged before this one: operands of the wave's two patches, 8 parts
        d2 pa0[2], pa1[2], pr0[2], pr1[2];
        bool pok[2] = {false, false};
        auto load_parts = [&](const double *buf, int ch) {
            const double *sr = buf + lo * TL::S;
#pragma unroll
            for (int pq = 0; pq < 2; ++pq) {
                const int p = 2 * wave + pq;
                const int gp = ch * TL::PATCHES + p;
                pok[pq] = p < TL::PATCHES && gp < a.np;
                const int pc = pok[pq] ? p : 0, gc = pok[pq] ? gp : 0;
                pa0[pq] = *reinterpret_cast<const d2 *>(sr + 16 * pc + 2 * k);
                pa1[pq] = *reinterpret_cast<const d2 *>(sr + 16 * pc + 8 + 2 * k);
                pr0[pq] = rt[128 * gc];
                pr1[pq] = rt[128 * gc + 64];
            }
        };
        auto part = [&](int e) {  // part e of 8: patch e >> 2, MFMA e & 3
            if (e >= 8) return;
            const int pq = e >> 2;
            if (!pok[pq]) return;
            const int m = e & 3;
            const double av = m == 0 ? pa0[pq].x : m == 1 ? pa0[pq].y : m == 2 ? pa1[pq].x : pa1[pq].y;
            const double rv = m == 0 ? pr0[pq].x : m == 1 ? pr0[pq].y : m == 2 ? pr1[pq].x : pr1[pq].y;
            acc = mfma_f64(av, rv, acc);
        };
        for (int ch = 0; ch < nch; ++ch) {
            double *buf = smem + (size_t)(ch & 1) * TL::BUF;
            const int rb0 = ch * MFB_RC_ADJ;
            const int nb = nrb - rb0 < MFB_RC_ADJ ? nrb - rb0 : MFB_RC_ADJ;
            // the next chunk's rows (the first chunk again after the last: the next tile starts there)
            of.fetch(g, (int64_t)(ch + 1 < nch ? rb0 + MFB_RC_ADJ : 0) * 64, tid);
            if (ch > 0) load_parts(smem + (size_t)((ch - 1) & 1) * TL::BUF, ch - 1);
            mfb_stage<KIND, TL::ROWS>(col, g, obs_s + oi * (MFB_NOBS * TL::ROWS), nb, lane, buf + wave * TL::S, nleaf,
                                      [&](int e) { if (ch > 0) part(e); });
            if (ch > 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (e >= nb) part(e);  // (fewer evaluations than parts: the rest here)
            }
            nent += (unsigned)nb;
            of.park(obs_s + (oi ^ 1) * (MFB_NOBS * TL::ROWS), tid);  // (last read while the previous chunk was staged)
            oi ^= 1;
            __syncthreads();
        }
        load_parts(smem + (size_t)((nch - 1) & 1) * TL::BUF, nch - 1);  // the tile's last chunk: nothing to hide behind
#pragma unroll
        for (int e = 0; e < 8; ++e) part(e);
        // the waves' accumulators through the staging buffer nobody reads any more, in wave order
        double *red = smem + (size_t)(nch & 1) * TL::BUF;
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave * 256 + q * 64 + lane] = acc[q];
        __syncthreads();
        if (tid < 256) {
            // acc[q] of lane (lo, k): column k + 4 q of the tile, chain lo
            const int cl = tid >> 4, c = tid & 15;
            const int ridx = (cl >> 2) * 64 + (cl & 3) * 16 + c;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < MFB_WAVES; ++w) s += red[w * 256 + ridx];
            const int64_t jc = tile * 16 + cl;
            if (jc < a.M) {
                const int64_t idx = jc * CB + c;
                if (Snear) s += Snear[idx];
                s = s * iw[jc];
                (void)mfb_update(a, idx, jc, c, 2.0 * s + (a.GREG ? a.GREG[idx] : 0.0), true, pp);
            }
        }
        __syncthreads();  // (the reduction buffer is the next tile's first staging buffer when nch is even)
    }
    if (tid < 256) ppred[tid >> 4][tid & 15] = pp;
    __syncthreads();
    if (a.pp_part && tid < 16 && (a.phase[tid] == PH_PFIN || a.phase[tid] == PH_PFIN_SPEC)) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += ppred[r][tid];
        a.pp_part[(int64_t)blockIdx.x * CB + tid] = t;
    }
    mfb_count(stats, nent, nleaf, lane);
}

// ---- forward of all chains --------------------------------------------------------------------
struct MfbFwdArgs {
    int64_t ld, M;
    const double *X;   // M x 16
    const double *iw;  // M: 1 / wm (1 where wm == 0 or the kernel is not weighted)
    int tiles_per_range;
    double *slab;      // gridDim.y x (ld x 16)
    int dbg;           // timing experiments only (GRAVHMC_MFB_DBG): 1 no barrier, 2 no MFMA, 4 no evaluation, 8 one column
};

template <int KIND>
__global__ void __launch_bounds__(1024)
mfb_forward_kernel(MfGeom g, MfbFwdArgs a, const double *__restrict__ cellc, MfStats *stats)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    using TL = MfbTile<MFB_RC_FWD>;
    double *obs_s = smem + 2 * TL::BUF;  // MFB_NOBS x ROWS: constants of the workgroup's rows, once
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, k = lane >> 4;
    const int64_t ntiles = (a.M + 15) / 16;
    const int nrb = (int)((a.ld + 63) / 64);
    const int rb0 = blockIdx.x * MFB_RC_FWD;
    const int nb = nrb - rb0 < MFB_RC_FWD ? nrb - rb0 : MFB_RC_FWD;
    const int64_t t0 = (int64_t)blockIdx.y * a.tiles_per_range;
    int64_t t1 = t0 + a.tiles_per_range;
    if (t1 > ntiles) t1 = ntiles;
    d4 acc[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
    unsigned nent = 0, nleaf = 0;
    {
        MfbObsFetch<KIND, TL::ROWS> of;
        of.fetch(g, (int64_t)rb0 * 64, tid);
        of.park(obs_s, tid);
    }
    __syncthreads();
    // MFMAs of the tile staged before this one: its XS fragments and 8 parts (4 column groups x 2 patches)
    double xsp[4] = {0.0, 0.0, 0.0, 0.0};
    const double *pbuf = smem;
    auto part = [&](int e) {
        if (e >= 8) return;
        const int u = e >> 1, pq = e & 1;
        const double av = pbuf[(4 * u + k) * TL::S + lo + 16 * (2 * wave + pq)];
        acc[pq] = mfma_f64(av, xsp[u], acc[pq]);
    };
    int it = 0;
    for (int64_t tile = t0; tile < t1; ++tile, ++it) {
        int64_t j = tile * 16 + wave;
        if (j >= a.M) j = a.M - 1;
        if (a.dbg & 8) j = wave;
        MfbCol<KIND> col;
        mfb_col_load<KIND>(col, g, cellc, j);
        double *buf = smem + (size_t)(it & 1) * TL::BUF;
        // this tile's XS fragments: lane (lo, k) feeds column 4 u + k, chain lo
        double xs[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t jj = tile * 16 + 4 * u + k;
            const bool ok = jj < a.M;
            const int64_t jc = ok ? jj : a.M - 1;
            const double v = a.X[jc * CB + lo] * a.iw[jc];
            xs[u] = ok ? v : 0.0;
        }
        if (a.dbg & 4) {
            for (int e = 0; e < nb; ++e) (buf + wave * TL::S)[e * 64 + lane] = obs_s[e * 64 + lane];
        } else {
            mfb_stage<KIND, TL::ROWS>(col, g, obs_s, nb, lane, buf + wave * TL::S, nleaf,
                                      [&](int e) { if (it > 0 && !(a.dbg & 2)) part(e); });
        }
        if (it > 0 && !(a.dbg & 2)) {
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (e >= nb) part(e);
        }
        nent += (unsigned)nb;
        if (!(a.dbg & 1)) __syncthreads();
        pbuf = buf;
#pragma unroll
        for (int u = 0; u < 4; ++u) xsp[u] = xs[u];
    }
    if (it > 0) {
#pragma unroll
        for (int e = 0; e < 8; ++e) part(e);  // the last tile: nothing to hide behind
    }
    // acc[pq][q] of lane (lo, k): row 16 (2 wave + pq) + k + 4 q of the chunk, chain lo
    double *out = a.slab + (int64_t)blockIdx.y * a.ld * CB;
#pragma unroll
    for (int pq = 0; pq < 2; ++pq)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = (int64_t)rb0 * 64 + 16 * (2 * wave + pq) + k + 4 * q;
            if (row < a.ld) out[row * CB + lo] = acc[pq][q];
        }
    mfb_count(stats, nent, nleaf, lane);
}

// ---- both products from ONE evaluation of every entry: teams of workgroups --------------------------
// The two-pass form above evaluates every entry twice per step because the update between the dot and
// the axpy needs the dot over ALL rows.  Here the workgroups that hold the row chunks of the same
// column tiles form a TEAM (grid = members x ranges of column tiles, every workgroup resident: one per
// CU) and hand each other their 16 x 16 partial dots per tile through memory, like the team sweep of
// the stored kernel (teamsweep.hip.h): a member stages its 448 rows of a tile once, forms its part of
// S from it, publishes the part, and -- one tile later, when everybody's parts have long arrived --
// sums the parts in member order, applies the leapfrog update (every member the same arithmetic, the
// same bits; member 0 stores the results) and feeds the new positions to the forward MFMAs on the
// tile it still holds in its other staging buffer.  One evaluation per entry and step for all chains.
//
// Exchange: the data is the flag (resident.hip.h): a double travels as two tagged 8-byte granules,
// written through, read with agent-scope loads; no counters, no fences; correct under any placement.
// A ring of four slots per team (a member is never more than one tile ahead of another).  Every wait
// is bounded (2 s): on a time-out the abort word is raised, every workgroup leaves, later launches
// return at once and the host repeats the work with the two-pass kernels.
constexpr int MFB_RC_FUS = 7;
constexpr int MFB_FUS_ADJW = 7;     // waves that run the adjoint MFMAs, four row patches each
constexpr int MFB_FUS_MAXMEM = 20;  // members of a team at most (N <= 8960 rows)
constexpr int MFB_FUS_RING = 4;
constexpr size_t MFB_LDS_FUS = (2 * (size_t)MfbTile<MFB_RC_FUS>::BUF + (size_t)MFB_NOBS * MfbTile<MFB_RC_FUS>::ROWS +
                                (size_t)MFB_FUS_ADJW * 256 + 4 * 256 + 256) * sizeof(double);  // 154112 B

struct MfbFusArgs {
    int tiles_per_range;
    double *slab;        // gridDim.y x (ld x 16)
    u64 *gran;           // [gridDim.y][MFB_FUS_RING][MFB_FUS_MAXMEM][256][2]
    unsigned tag0;       // tags tag0 + 1 .. tag0 + tiles_per_range belong to this launch
    unsigned *abort_w;
    int poll_members;    // gridDim.x (+ 1 in the time-out test: one part never comes)
    int n_pp;            // rows of pp_part the host sums (those beyond the ranges are zeroed)
    long long *dbg;      // optional: 8 accumulated phase times (100 MHz ticks) of workgroup (0, 0), wave 0
};

template <int KIND>
__global__ void __launch_bounds__(1024)
mfb_fused_kernel(MfGeom g, BatchAdjArgs a, MfbFusArgs f, const double *__restrict__ iw, const double *__restrict__ cellc,
                 const double *__restrict__ Snear, MfStats *stats)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double ppred[16][17];
    __shared__ double chs[3][16];  // phase, cu, cp of the chains
    __shared__ long long tph_s[9];
    __shared__ int abort_s;
    using TL = MfbTile<MFB_RC_FUS>;
    double *obs_s = smem + 2 * TL::BUF;                  // MFB_NOBS x ROWS
    double *red = obs_s + MFB_NOBS * TL::ROWS;           // MFB_FUS_ADJW x 256: the adjoint waves' accumulators
    double *gp = red + MFB_FUS_ADJW * 256;               // 2 x 256: sums over the members q, q + 2, ...
    double *xs_s = gp + 4 * 256;                         // 256: XS of the tile being finished, [col][chain]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, k = lane >> 4;
    const int mem = blockIdx.x, cr = blockIdx.y;
    const int64_t ntiles = (a.M + 15) / 16;
    const int nrb = (int)((a.ld + 63) / 64);
    const int rb0 = mem * MFB_RC_FUS;
    const int nb = nrb - rb0 < MFB_RC_FUS ? nrb - rb0 : MFB_RC_FUS;
    const int64_t t0 = (int64_t)cr * f.tiles_per_range;
    const int ntl = (int)((ntiles - t0 < f.tiles_per_range) ? ntiles - t0 : f.tiles_per_range);
    if (tid == 0) abort_s = (__hip_atomic_load(f.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
    {
        MfbObsFetch<KIND, TL::ROWS> of;
        of.fetch(g, (int64_t)rb0 * 64, tid);
        of.park(obs_s, tid);
    }
    // adjoint: wave w < 7 contracts the row patches w, w + 7, w + 14, w + 21 of the member
    const d2 *rt = reinterpret_cast<const d2 *>(a.Rt) + (k * 16 + lo);
    bool aok[4];
    int agp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int p = wave + MFB_FUS_ADJW * q;
        const int gpi = mem * TL::PATCHES + p;
        aok[q] = wave < MFB_FUS_ADJW && p < TL::PATCHES && gpi < a.np;
        agp[q] = aok[q] ? gpi : 0;
    }
    u64 *gteam = f.gran + (size_t)cr * MFB_FUS_RING * MFB_FUS_MAXMEM * 512;
    auto gran_of = [&](int it, int member) -> u64 * {
        return gteam + ((size_t)(it & (MFB_FUS_RING - 1)) * MFB_FUS_MAXMEM + member) * 512;
    };
    d4 accf[2] = {d4{0.0, 0.0, 0.0, 0.0}, d4{0.0, 0.0, 0.0, 0.0}};
    double pp = 0.0;
    unsigned nent = 0, nleaf = 0;
    // the chain of this thread in the exchange / the update never changes: its phase and coefficients once
    // (parked in LDS: as registers they push the evaluation loop into spills)
    if (tid < 16) {
        chs[0][tid] = (double)a.phase[tid];
        chs[1][tid] = a.cu[tid];
        chs[2][tid] = a.cp[tid];
    }
    const bool my_live = a.phase[tid & 15] != PH_IDLE;  // (idle chains -- and the unused slots of a batch of < 16 -- exchange nothing)
    __syncthreads();
    if (abort_s) return;  // an earlier launch of this stream gave up: the host repeats the work
    // (per-phase clocks of one thread, kept in LDS: as registers they cost every thread 18 VGPRs)
    const bool clk = f.dbg != nullptr && mem == 0 && cr == 0 && tid == 0;
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
    for (int it = 0; it <= ntl; ++it) {
        const bool stg = it < ntl, fin = it >= 1;
        double *buf = smem + (size_t)(it & 1) * TL::BUF;
        if (stg) {
            int64_t j = (t0 + it) * 16 + wave;
            if (j >= a.M) j = a.M - 1;
            MfbCol<KIND> col;
            mfb_col_load<KIND>(col, g, cellc, j);
            mfb_stage<KIND, TL::ROWS>(col, g, obs_s, nb, lane, buf + wave * TL::S, nleaf, [](int) {});
            nent += (unsigned)nb;
        }
        mark(0);
        __syncthreads();  // the tile is staged; everybody is done with the forward of tile it - 2
        mark(1);
        // Waves 0 .. 6: this member's part of S for the tile just staged (adjoint MFMAs).  Waves 8 .. 15
        // meanwhile: the parts of tile it - 1, published an iteration ago -- thread (v, q) collects the
        // members q, q + 2, ...: every load requested before anything is looked at, from uniform bases at
        // one lane offset; waves 8 .. 11 also request the operands of the update of their (cell, chain).
        const int gv = tid & 255;
        const int gq = __builtin_amdgcn_readfirstlane((tid >> 8) & 1);
        const unsigned gtag = f.tag0 + (unsigned)it;  // tile it - 1 carries tag0 + (it - 1) + 1
        MfbUpd up;
        const int ucl = gv >> 4, uc = gv & 15;
        const int64_t ujc = (t0 + it - 1) * 16 + ucl;
        const bool uthr = wave >= 8 && wave < 12;      // threads 512 .. 767: one per (cell, chain) of the tile
        const bool uok = fin && uthr && ujc < a.M;
        if (stg && wave < MFB_FUS_ADJW) {
            d2 ar0[4], ar1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                ar0[q] = rt[128 * agp[q]];
                ar1[q] = rt[128 * agp[q] + 64];
            }
            const double *sr = buf + lo * TL::S;
            d4 acc = d4{0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (aok[q]) {
                    const int p = wave + MFB_FUS_ADJW * q;
                    const d2 a0 = *reinterpret_cast<const d2 *>(sr + 16 * p + 2 * k);
                    const d2 a1 = *reinterpret_cast<const d2 *>(sr + 16 * p + 8 + 2 * k);
                    acc = mfma_f64(a0.x, ar0[q].x, acc);
                    acc = mfma_f64(a0.y, ar0[q].y, acc);
                    acc = mfma_f64(a1.x, ar1[q].x, acc);
                    acc = mfma_f64(a1.y, ar1[q].y, acc);
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave * 256 + q * 64 + lane] = acc[q];
        }
        if (fin && wave >= 8) {
            // (one 16-byte `global_load_dwordx4 sc0 sc1` per double instead of two 8-byte agent-scope loads
            // was measured slower: 3.61 against 3.22 ms per pass)
            u64 ga[MFB_FUS_MAXMEM / 2], gb[MFB_FUS_MAXMEM / 2];
#pragma unroll
            for (int i = 0; i < MFB_FUS_MAXMEM / 2; ++i) {
                const int m = gq + 2 * i;
                ga[i] = gb[i] = 0;
                if (m < f.poll_members && my_live) ld_gran_issue(gran_of(it - 1, m) + 2 * gv, ga[i], gb[i]);
            }
            if (uok) mfb_upd_load(up, a, Snear, iw, ujc * CB + uc, ujc);
            double sum = 0.0;
            bool ok = true;
#pragma unroll
            for (int i = 0; i < MFB_FUS_MAXMEM / 2; ++i) {
                const int m = gq + 2 * i;
                if (m < f.poll_members && my_live) {
                    double val = 0.0;
                    if (!gran_value(ga[i], gb[i], gtag, val)) {
                        // (a member lags by more than an iteration) poll until its part is there
                        unsigned spins = 0;
                        long long tstart = 0;
                        while (ok && !ld_gran(gran_of(it - 1, m) + 2 * gv, gtag, val)) {
                            __builtin_amdgcn_s_sleep(1);
                            if ((++spins & 63u) == 0) {
                                const long long now = wall_clock64();
                                if (tstart == 0) tstart = now;
                                if (__hip_atomic_load(f.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                                    now - tstart > RES_TIMEOUT_TICKS) {
                                    __hip_atomic_store(f.abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = false;
                                }
                            }
                        }
                    }
                    sum += val;
                }
            }
            gp[gq * 256 + gv] = sum;
            if (!ok) abort_s = 1;
        }
        mark(2);
        __syncthreads();
        mark(3);
        if (stg && tid < 256) {
            // this member's part of S for tile it (acc[q] of lane (lo, k): column k + 4 q, chain lo)
            const int cl = tid >> 4, c = tid & 15;
            const int ridx = (cl >> 2) * 64 + (cl & 3) * 16 + c;
            if (my_live) {
                double sp = 0.0;
#pragma unroll
                for (int w = 0; w < MFB_FUS_ADJW; ++w) sp += red[w * 256 + ridx];
                st_gran(gran_of(it, mem) + 2 * tid, f.tag0 + (unsigned)it + 1u, sp);
            }
        }
        if (fin && uthr) {
            double xs = 0.0;
            if (uok) {
                double st = gp[gv] + gp[256 + gv];
                st = (st + up.snear) * up.iw;
                const double xn = mfb_update_pre(a, up, ujc * CB + uc, (int)chs[0][uc], chs[1][uc], chs[2][uc],
                                                 2.0 * st + up.greg, mem == 0, pp);
                xs = xn * up.iw;
            }
            xs_s[gv] = xs;
        }
        mark(4);
        __syncthreads();
        mark(5);
        if (abort_s) return;
        if (fin && wave < TL::PATCHES / 2) {
            const double *pbuf = smem + (size_t)((it - 1) & 1) * TL::BUF;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double b = xs_s[(4 * u + k) * 16 + lo];
                const double *sa = pbuf + (4 * u + k) * TL::S + lo;
#pragma unroll
                for (int pq = 0; pq < 2; ++pq) accf[pq] = mfma_f64(sa[16 * (2 * wave + pq)], b, accf[pq]);
            }
        }
        mark(6);
        __syncthreads();  // (the next tile is staged into the buffer the forward just read)
        mark(7);
    }
    if (clk)
        for (int q = 0; q < 8; ++q) f.dbg[q] += tph_s[q];
    double *out = f.slab + (int64_t)cr * a.ld * CB;
    if (wave < TL::PATCHES / 2) {
#pragma unroll
        for (int pq = 0; pq < 2; ++pq)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t row = (int64_t)rb0 * 64 + 16 * (2 * wave + pq) + k + 4 * q;
                if (row < a.ld) out[row * CB + lo] = accf[pq][q];
            }
    }
    if (mem == 0 && a.pp_part) {
        if (tid >= 512 && tid < 768) ppred[(tid - 512) >> 4][tid & 15] = pp;
        __syncthreads();
        if (tid < 16 && (a.phase[tid] == PH_PFIN || a.phase[tid] == PH_PFIN_SPEC)) {
            double t = 0.0;
#pragma unroll
            for (int r = 0; r < 16; ++r) t += ppred[r][tid];
            a.pp_part[(int64_t)cr * CB + tid] = t;
            // (the host sums n_pp rows: the two-pass adjoint writes more of them)
            if (cr == 0)
                for (int r = (int)gridDim.y; r < f.n_pp; ++r) a.pp_part[(int64_t)r * CB + tid] = 0.0;
        }
    }
    mfb_count(stats, nent, nleaf, lane);
}

// ---- ONE chain on teams: the single-chain matrix-free pass without a column's phases ---------------
// mf_tess_fast_kernel (kernels.hip.h) walks a column in lock-step phases inside one workgroup -- constants,
// 8 slots, barrier, scalars, update, forward -- and issues VALU 80 % of the time at 127 instructions per
// entry.  The staging phase above runs at 97 instructions per entry with the VALU saturated, and a single
// chain needs neither MFMAs nor the staged tile's transposition: wave w keeps its column's dot with r as
// ONE number per tile, so a team's exchange is 16 doubles per member and tile.  Same grid and exchange as
// mfb_fused_kernel (members x ranges, every workgroup resident; the data is the flag), one tile of lag:
//   tile it:      evaluate (values to LDS, dot with the member's rows of r on the fly); in the middle of
//                 the evaluations four waves request the team's parts of tile it - 1 and the operands of
//                 its update, and right behind their own evaluations sum the parts (row scan over the
//                 members' lanes) and apply the leapfrog update (hmc.py:114-152): x / wm to LDS
//   barrier       (the only one per tile) publish the 16 partial dots of tile it; every wave: forward of
//                 ITS column of tile it - 1 from the values it parked in LDS
// The forward partials of the 16 waves meet once, at the end of the launch.  Near-field pairs as in the
// batch (differences through two small sparse kernels).  Modes of SweepArgs with SW_ADJ.
constexpr int MFT_MAXMEM = 40;  // members of a team at most (N <= 17920 rows)
constexpr size_t MFT_LDS = (2 * (size_t)MfbTile<MFB_RC_FUS>::BUF + (size_t)MFB_NOBS * MfbTile<MFB_RC_FUS>::ROWS +
                            (size_t)MfbTile<MFB_RC_FUS>::ROWS + 32 + 32 + 16) * sizeof(double);

struct MftArgs {
    int tiles_per_range;
    u64 *gran;           // [gridDim.y][MFB_FUS_RING][MFT_MAXMEM][16][2]
    unsigned tag0;
    unsigned *abort_w;
    int poll_members;    // gridDim.x (+ 1 in the time-out test)
    int n_pp;            // entries of pp_part the host sums
    const double *snear; // M: near-field part of the dots, or nullptr
};

template <int KIND>
__global__ void __launch_bounds__(1024)
mf_team_kernel(MfGeom g, SweepArgs a, MftArgs f, const double *__restrict__ wm, const double *__restrict__ cellc,
               MfStats *stats)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int abort_s;
    using TL = MfbTile<MFB_RC_FUS>;
    double *obs_s = smem + 2 * TL::BUF;              // MFB_NOBS x ROWS
    double *r_s = obs_s + MFB_NOBS * TL::ROWS;       // ROWS: the member's rows of r
    double *part = r_s + TL::ROWS;                   // 2 x 16: the waves' dots of the tile just evaluated (by parity)
    double *xs_s = part + 32;                        // 2 x 16: x / wm of the tile being finished (by parity)
    double *ppred = xs_s + 32;                       // 16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mem = blockIdx.x, cr = blockIdx.y;
    const int mode = a.mode;
    const int64_t ntiles = (a.M + 15) / 16;
    const int nrb = (int)((a.ld + 63) / 64);
    const int rb0 = mem * MFB_RC_FUS;
    const int nb = nrb - rb0 < MFB_RC_FUS ? nrb - rb0 : MFB_RC_FUS;
    const int64_t t0 = (int64_t)cr * f.tiles_per_range;
    const int ntl = (int)((ntiles - t0 < f.tiles_per_range) ? ntiles - t0 : f.tiles_per_range);
    if (tid == 0) abort_s = (__hip_atomic_load(f.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
    {
        MfbObsFetch<KIND, TL::ROWS> of;
        of.fetch(g, (int64_t)rb0 * 64, tid);
        of.park(obs_s, tid);
        if (tid < TL::ROWS) {
            const int64_t row = (int64_t)rb0 * 64 + tid;
            r_s[tid] = row < g.N ? a.r[row] : 0.0;
        }
    }
    u64 *gteam = f.gran + (size_t)cr * MFB_FUS_RING * MFT_MAXMEM * 32;
    auto gran_of = [&](int it, int member) -> u64 * {
        return gteam + ((size_t)(it & (MFB_FUS_RING - 1)) * MFT_MAXMEM + member) * 32;
    };
    double dacc[MFB_RC_FUS];
#pragma unroll
    for (int e = 0; e < MFB_RC_FUS; ++e) dacc[e] = 0.0;
    double pp = 0.0;
    unsigned nent = 0, nleaf = 0;
    // Exchange roles (waves 0 .. 3 only; they finish their evaluations first): lane l of wave v collects, for
    // column 4 v + (l >> 4) of the tile, the parts of the members (l & 15), (l & 15) + 16, (l & 15) + 32; a row
    // scan adds the 16 lanes of a column (fixed order), and the row's last lane applies the update.  No LDS,
    // no barrier between collecting and updating.
    constexpr int GS = (MFT_MAXMEM + 15) / 16;
    const bool gwave = wave < 4;
    const int gcol = 4 * wave + (lane >> 4), gm0 = lane & 15;
    const bool ulane = gwave && gm0 == 15;
    __syncthreads();
    if (abort_s) return;
    for (int it = 0; it <= ntl; ++it) {
        const bool stg = it < ntl, fin = it >= 1;
        double *buf = smem + (size_t)(it & 1) * TL::BUF;
        const unsigned gtag = f.tag0 + (unsigned)it;  // tile it - 1 carries tag0 + (it - 1) + 1
        u64 ga[GS], gb[GS];
        double u_w = 1.0, u_x = 0.0, u_g = 0.0, u_p = 0.0, u_pn = 0.0, u_hi = 0.0, u_lo = 0.0, u_sn = 0.0;
        const int64_t uj = (t0 + it - 1) * 16 + gcol;  // the update lane's cell: (tile it - 1, column gcol)
        const bool uok = fin && ulane && uj < a.M;
        auto request = [&]() {
            // (issued in the middle of the evaluations: the parts were published an evaluation phase ago and
            // have mostly arrived by the time they are looked at -- their round trip hides behind the rest)
            if (fin && gwave) {
#pragma unroll
                for (int q = 0; q < GS; ++q) {
                    ga[q] = gb[q] = 0;
                    if (gm0 + 16 * q < f.poll_members) ld_gran_issue(gran_of(it - 1, gm0 + 16 * q) + 2 * gcol, ga[q], gb[q]);
                }
            }
            if (uok) {
                u_w = wm ? wm[uj] : 1.0;
                u_x = (mode & (SW_UPD | SW_FWD)) ? a.x_in[uj] : 0.0;  // (a final half step alone carries no position)
                u_g = a.greg ? a.greg[uj] : 0.0;
                if (mode & (SW_PFIN | SW_UPD)) u_p = a.p_in[uj];
                if (mode & SW_SPEC) u_pn = a.pn_in[uj];
                if (mode & SW_UPD) {
                    u_hi = a.high[uj];
                    u_lo = a.low[uj];
                }
                u_sn = f.snear ? f.snear[uj] : 0.0;
            }
        };
        if (stg) {
            int64_t j = (t0 + it) * 16 + wave;
            if (j >= a.M) j = a.M - 1;
            MfbCol<KIND> col;
            mfb_col_load<KIND>(col, g, cellc, j);
            double *st = buf + wave * TL::S;
            double s = 0.0;
            // (a copy of mfb_stage with the dot folded in and the requests in the middle)
            if (nb == MFB_RC_FUS && KIND >= 2) {
#pragma unroll
                for (int e = 0; e < MFB_RC_FUS; ++e) {
                    double o[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) o[q] = q < mfb_nobs<KIND>() ? obs_s[q * TL::ROWS + e * 64 + lane] : 0.0;
                    const double v = mfb_eval<KIND>(col, g, o, nleaf);
                    st[e * 64 + lane] = v;
                    s = fma(v, r_s[e * 64 + lane], s);
                    if (e == MFB_RC_FUS / 2) request();
                }
            } else {
#pragma unroll 1
                for (int e = 0; e < nb; ++e) {
                    double o[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) o[q] = q < mfb_nobs<KIND>() ? obs_s[q * TL::ROWS + e * 64 + lane] : 0.0;
                    const double v = mfb_eval<KIND>(col, g, o, nleaf);
                    st[e * 64 + lane] = v;
                    s = fma(v, r_s[e * 64 + lane], s);
                }
                request();
            }
            nent += (unsigned)nb;
            s = wave_sum_dpp(s);
            if (lane == 0) part[(it & 1) * 16 + wave] = s;
        } else {
            request();
        }
        if (fin && gwave) {
            // the team's parts of tile it - 1 (poll what has not arrived), summed over the members
            double sum = 0.0;
            bool ok = true;
#pragma unroll
            for (int q = 0; q < GS; ++q) {
                const int m = gm0 + 16 * q;
                if (m < f.poll_members) {
                    double val = 0.0;
                    if (!gran_value(ga[q], gb[q], gtag, val)) {
                        unsigned spins = 0;
                        long long tstart = 0;
                        while (ok && !ld_gran(gran_of(it - 1, m) + 2 * gcol, gtag, val)) {
                            __builtin_amdgcn_s_sleep(1);
                            if ((++spins & 63u) == 0) {
                                const long long now = wall_clock64();
                                if (tstart == 0) tstart = now;
                                if (__hip_atomic_load(f.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                                    now - tstart > RES_TIMEOUT_TICKS) {
                                    __hip_atomic_store(f.abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = false;
                                }
                            }
                        }
                    }
                    sum += val;
                }
            }
            if (!ok) abort_s = 1;
            const double tot = row16_sum_dpp(sum);  // lane 15 of every row of 16: the column's dot over all members
            if (ulane) {
                double xs = 0.0;
                if (uok) {
                    const double iwj = (u_w != 0.0) ? 1.0 / u_w : 1.0;
                    const double t = (tot + u_sn) * iwj;
                    const double grad = 2.0 * t + u_g;
                    const bool wr = mem == 0;
                    double xj = u_x;
                    if ((mode & SW_GOUT) && wr) a.g_out[uj] = grad;
                    if (mode & SW_PFIN) {
                        const double pf = u_p - a.c_p * grad;
                        if (wr) pp += pf * pf;
                        if (!(mode & SW_SPEC) && wr) a.p_out[uj] = pf;
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
                        if (wr) {
                            a.p_out[uj] = pj;
                            a.x_out[uj] = xj;
                        }
                    }
                    xs = xj * iwj;
                }
                xs_s[(it & 1) * 16 + gcol] = xs;
            }
        }
        __syncthreads();  // the tile is staged, its 16 dots are in `part`, the finished tile's x / wm in `xs_s`
        if (abort_s) return;
        if (stg && tid < 16) st_gran(gran_of(it, mem) + 2 * tid, f.tag0 + (unsigned)it + 1u, part[(it & 1) * 16 + tid]);
        if (fin && (mode & SW_FWD)) {
            const double *pv = smem + (size_t)((it - 1) & 1) * TL::BUF + wave * TL::S;
            const double xw = xs_s[(it & 1) * 16 + wave];
#pragma unroll
            for (int e = 0; e < MFB_RC_FUS; ++e) dacc[e] = fma(pv[e * 64 + lane], xw, dacc[e]);
        }
        // (ONE barrier per tile: a wave reads and re-stages only its own column of a staging buffer, and
        // `part` / `xs_s` alternate by the tile's parity -- nobody is more than a barrier behind)
    }
    if (mode & SW_FWD) {
        // the 16 waves' partials of the member's rows, in wave order
        __syncthreads();
        double *sum_s = smem;  // 16 x ROWS (the staging buffers are free)
#pragma unroll
        for (int e = 0; e < MFB_RC_FUS; ++e) sum_s[wave * TL::ROWS + e * 64 + lane] = dacc[e];
        __syncthreads();
        double *out = a.slab + (int64_t)cr * a.ld;
        if (tid < TL::ROWS) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < MFB_WAVES; ++w) t += sum_s[w * TL::ROWS + tid];
            const int64_t row = (int64_t)rb0 * 64 + tid;
            if (tid < nb * 64 && row < a.ld) out[row] = t;
        }
    }
    if ((mode & SW_PFIN) && mem == 0) {
        __syncthreads();
        if (ulane) ppred[gcol] = pp;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int q = 0; q < 16; ++q) t += ppred[q];
            a.pp_part[cr] = t;
            if (cr == 0)
                for (int q = (int)gridDim.y; q < f.n_pp; ++q) a.pp_part[q] = 0.0;
        }
    }
    mfb_count(stats, nent, nleaf, lane);
}

// near-field differences for ONE chain: snear[j] = sum over the listed rows of column j of delta * r[row]
__global__ void __launch_bounds__(64)
mf1_near_adjoint_kernel(const int64_t *__restrict__ ptr, const int *__restrict__ row, const double *__restrict__ delta,
                        const double *__restrict__ r, double *__restrict__ snear)
{
    const int64_t j = blockIdx.x;
    const int64_t q0 = ptr[j], q1 = ptr[j + 1];
    double s = 0.0;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += 64) s += delta[q] * r[row[q]];
    s = wave_sum_dpp(s);
    if (threadIdx.x == 0) snear[j] = s;
}

// out[i] = sum over the listed columns of row i of delta * x[col] / wm[col]: one more row of the slab
__global__ void __launch_bounds__(64)
mf1_near_forward_kernel(const int64_t *__restrict__ rptr, const int *__restrict__ rcol, const double *__restrict__ rdelta,
                        int64_t N, const double *__restrict__ x, const double *__restrict__ wm, double *__restrict__ out)
{
    const int64_t i = blockIdx.x;
    double s = 0.0;
    if (i < N)
        for (int64_t q = rptr[i] + threadIdx.x; q < rptr[i + 1]; q += 64) {
            const int64_t j = rcol[q];
            const double w = wm ? wm[j] : 1.0;
            s += rdelta[q] * ((w != 0.0) ? x[j] * (1.0 / w) : x[j]);
        }
    s = wave_sum_dpp(s);
    if (threadIdx.x == 0) out[i] = s;
}

// ---- near-field list as differences ---------------------------------------------------------------
// delta[q] = val[q] - (the root leaf the dense passes stage for that pair), column-major order
template <int KIND>
__global__ void __launch_bounds__(256)
mfb_near_delta_kernel(MfGeom g, const double *__restrict__ cellc, MfNear near, int64_t n, const int *__restrict__ colof,
                      double *__restrict__ delta)
{
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= n) return;
    const int64_t i = near.row[q], j = colof[q];
    const double *cc = cellc + (int64_t)TESS_NC * j;
    double leaf;
    if constexpr (KIND == 3)
        leaf = tess_leaf_fast(g.o4[i], g.o5[i], g.o1[i], g.o2[i], g.o3[i], cc);
    else
        leaf = tess_leaf_cc(g.o0[i], g.o1[i], g.o2[i], g.o3[i], cc);
    delta[q] = near.val[q] - leaf;
}

// offset of residual (row i, chain c) in the patch-transposed layout of batch.hip.h
__device__ __forceinline__ int64_t rt_offset(int64_t i, int c)
{
    const int64_t p = i >> 4, q = i & 15;
    return (((p * 2 + (q >> 3)) * 4 + ((q & 7) >> 1)) * CB + c) * 2 + (q & 1);
}

// Snear[j][c] = sum over the listed rows of column j of delta * r_c[row].  One workgroup per column:
// thread (e, c) takes the entries e, e + 16, ... of chain c; the 16 partial sums are added in the order
// of e (fixed: reproducible).  (A thread per (column, chain) walking its list alone cost 70 us per
// pass at C4: the polar columns list hundreds of rows.)
__global__ void __launch_bounds__(256)
mfb_near_adjoint_kernel(const int64_t *__restrict__ ptr, const int *__restrict__ row, const double *__restrict__ delta,
                        int64_t M, const double *__restrict__ Rt, double *__restrict__ Snear)
{
    __shared__ double part[16][17];
    const int64_t j = blockIdx.x;
    const int c = threadIdx.x & 15, e = threadIdx.x >> 4;
    const int64_t q0 = ptr[j], q1 = ptr[j + 1];
    if (q1 == q0) {  // (uniform)
        if (e == 0) Snear[j * CB + c] = 0.0;
        return;
    }
    double s = 0.0;
    for (int64_t q = q0 + e; q < q1; q += 16) s += delta[q] * Rt[rt_offset(row[q], c)];
    part[e][c] = s;
    __syncthreads();
    if (e == 0) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += part[r][c];
        Snear[j * CB + c] = t;
    }
}

// out[i][c] = sum over the listed columns of row i of delta * XS[col][c]; one more block of the
// forward slab.  One workgroup per row, as above.
__global__ void __launch_bounds__(256)
mfb_near_forward_kernel(const int64_t *__restrict__ rptr, const int *__restrict__ rcol, const double *__restrict__ rdelta,
                        int64_t N, int64_t ld, const double *__restrict__ X, const double *__restrict__ iw,
                        double *__restrict__ out)
{
    __shared__ double part[16][17];
    const int64_t i = blockIdx.x;
    const int c = threadIdx.x & 15, e = threadIdx.x >> 4;
    const int64_t q0 = i < N ? rptr[i] : 0, q1 = i < N ? rptr[i + 1] : 0;
    if (q1 == q0) {
        if (e == 0) out[i * CB + c] = 0.0;
        return;
    }
    double s = 0.0;
    for (int64_t q = q0 + e; q < q1; q += 16) {
        const int64_t j = rcol[q];
        s += rdelta[q] * (X[j * CB + c] * iw[j]);
    }
    part[e][c] = s;
    __syncthreads();
    if (e == 0) {
        double t = 0.0;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += part[r][c];
        out[i * CB + c] = t;
    }
}

__global__ void __launch_bounds__(256) mfb_invw_kernel(const double *wm, int64_t M, double *iw)
{
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= M) return;
    const double w = wm ? wm[j] : 1.0;
    iw[j] = (w != 0.0) ? 1.0 / w : 1.0;
}

}  // namespace ghk
