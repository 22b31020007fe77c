// Team sweep (gfx950): the fused one-read sweep of kernels.hip.h for more observations than ONE
// workgroup can hold of a column (N > 16384: BASELINE configs[4], 200 x 200 observations).
//
// sweep_kernel keeps a column in the registers of one workgroup between the dot with r and the
// axpy into d -- that is what lets the adjoint of step s, the leapfrog update and the forward of
// step s+1 share ONE read of G (potential.py:698,708 are two).  A column of 4*10^4 rows is 320 KB:
// more than a workgroup's registers.  Round 1 therefore ran such problems in row panels (adjoint of
// every panel, elementwise update, forward of every panel: two reads of G per step).  Here a TEAM of
// Q = 2..8 workgroups shares the column: member q keeps rows [q R, (q+1) R) of it in registers
// (R <= 10240: five double2 per thread), forms its part of the dot and publishes it; every member
// adds the Q parts in the same order, applies the same update and adds its rows times the new x_j to
// its part of the forward partial.  One read of G again.
//
// The exchange is the resident chain kernel's (resident.hip.h): the data is the flag -- a double
// travels as two 8-byte granules {tag, 32 bits}, each written by one write-through store; no
// counters, no fences.  Two things keep it off the critical path:
//   * a lag of one column: a member publishes its part of column i and only then collects the parts
//     of column i-1 -- published a whole column (~3 us of HBM streaming) earlier -- finishes that
//     column, whose registers it kept meanwhile, and requests the column after next into them
//     (three column buffers rotate: finishing | dotted | in flight);
//   * the parts are polled with ONE scalar load (s_load_dwordx16 glc: the Q granule pairs of a ring
//     slot are contiguous).  Scalar memory operations are counted by lgkmcnt, not vmcnt: the poll
//     does not queue behind the wave's column loads in flight.  (A vector poll returns only after
//     every older vector load of its wave: measured, the team then streams one column at a time,
//     5.2 instead of 6 TB/s.)  A scalar load reads the XCD's L2, so the members of a team must share
//     an XCD: they are blocks with equal blockIdx % 8, and the host checks once with a probe launch
//     that the dispatcher places such blocks on one XCD (else: row panels).
// A ring of four slots suffices (no member is ever more than two columns ahead of another of its
// team).  Every wait is bounded (2 s): on a time-out the abort word is raised, every workgroup
// leaves, later launches of the stream return at once and the host repeats the work in row panels
// (host_sweep.h: team_failed).
#pragma once

namespace ghk {

constexpr int TS_MAXQ = 8;
constexpr int TS_RING = 4;
constexpr int TS_MAXWAVES = 16;

// The granule pairs of up to 8 members (16 bytes each, contiguous) with scalar loads that bypass the
// scalar cache (glc) and read the XCD's L2, where a same-XCD writer's write-through store has just
// passed.  ok: every one of the first n pairs carries `tag`; sum: their values added in member
// order (the first nsum of them).
typedef unsigned ts_u16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ bool poll_parts_scalar(const u64 *g, int n, int nsum, unsigned tag, double &sum)
{
    ts_u16 lo, hi;
    if (n > 4) {
        asm volatile("s_load_dwordx16 %0, %2, 0x0 glc\n\ts_load_dwordx16 %1, %2, 0x40 glc\n\ts_waitcnt lgkmcnt(0)"
                     : "=&s"(lo), "=&s"(hi)
                     : "s"(g)
                     : "memory");
    } else {
        asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(lo) : "s"(g) : "memory");
        hi = lo;
    }
    bool ok = true;
    double t = 0.0;
#pragma unroll
    for (int m = 0; m < TS_MAXQ; ++m) {
        if (m < n) {
            const ts_u16 &w = m < 4 ? lo : hi;
            const int o = 4 * (m & 3);
            ok = ok && w[o + 1] == tag && w[o + 3] == tag;
            if (m < nsum) t += __longlong_as_double((long long)(((u64)w[o + 2] << 32) | (u64)w[o]));
        }
    }
    sum = t;
    return ok;
}

struct TeamArgs {
    SweepArgs s;          // mode, G, ld, M, vectors, coefficients (row0 / rows / n_teams unused)
    int Q;                // members per team
    int poll_q;           // parts a member waits for (= Q; Q + 1 in the time-out test: one never comes)
    int tpx;              // teams per XCD slot (blockIdx % 8): grid = 8 * tpx * Q
    int64_t panel_rows;   // rows per member (multiple of 16, <= 10240)
    int64_t cols_per_team;
    int n_pp;             // entries of pp_part the host sums (those beyond the teams are zeroed)
    u64 *gran;            // [8 * tpx][TS_RING][TS_MAXQ][2]
    unsigned tag0;        // tags tag0 + 1 .. tag0 + cols_per_team belong to this launch
    unsigned *abort_w;
};

// TS_THREADS threads per workgroup, TS_EPT2 double2 per thread and column (rows per member <=
// TS_THREADS * TS_EPT2 * 2), TS_D column buffers (>= 3): finishing | dotted | TS_D - 2 in flight.
template <int TS_THREADS, int TS_EPT2, int TS_D>
__global__ void __launch_bounds__(TS_THREADS) teamsweep_kernel(TeamArgs a)
{
    constexpr int TS_WAVES = TS_THREADS / 64;
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const SweepArgs &s = a.s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int mode = s.mode;
    const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
    const int q = y % a.Q, team = (y / a.Q) * 8 + x;
    const int64_t ld = s.ld;
    const int64_t row0 = (int64_t)q * a.panel_rows;
    const int64_t rows = (row0 + a.panel_rows <= ld) ? a.panel_rows : (ld > row0 ? ld - row0 : 0);
    const int ld2 = (int)(rows >> 1);
    // LDS: r of the member's rows | dot partials, 2 x 16 | per-column scalars, ring of 4 x 8 | total, flag
    double *r_s = smem;
    double *part = smem + a.panel_rows;
    double *scal = part + 2 * TS_MAXWAVES;
    double *tot_s = scal + TS_RING * 8;
    int *abort_s = reinterpret_cast<int *>(tot_s + 2);

    if (tid == 0) *abort_s = (__hip_atomic_load(a.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
    {
        const d2 *r2 = reinterpret_cast<const d2 *>(s.r + row0);
        d2 *rs2 = reinterpret_cast<d2 *>(r_s);
        for (int e = tid; e < ld2; e += TS_THREADS) rs2[e] = r2[e];
    }
    __syncthreads();
    if (*abort_s) return;  // an earlier launch of this stream gave up: the host repeats the work

    const int64_t jb = (int64_t)team * a.cols_per_team;
    int64_t jend = jb + a.cols_per_team;
    if (jend > s.M) jend = s.M;
    const int cnt = jb < jend ? (int)(jend - jb) : 0;
    u64 *gteam = a.gran + (size_t)team * TS_MAXQ * TS_RING * 2;

    d2 dacc[TS_EPT2];
#pragma unroll
    for (int k = 0; k < TS_EPT2; ++k) dacc[k] = d2{0.0, 0.0};
    double pp = 0.0;

    auto load_col = [&](ColRegs<TS_EPT2> &c, int i) {
        const int64_t j = jb + i;
        const d2 *col = reinterpret_cast<const d2 *>(s.G + j * ld + row0);
#pragma unroll
        for (int k = 0; k < TS_EPT2; ++k) {
            const int e = k * TS_THREADS + tid;
            c.v[k] = (e < ld2) ? __builtin_nontemporal_load(col + e) : d2{0.0, 0.0};
        }
        double sc = 0.0;
        if (wave == 0 && lane < 6) {
            const double *src = lane == 0 ? s.x_in : lane == 1 ? s.p_in : lane == 2 ? s.low
                              : lane == 3 ? s.high : lane == 4 ? s.greg : s.pn_in;
            if (src) sc = src[j];
        }
        c.sc = sc;
    };

    // B(i): the member's part of <G_j, r>, published for the team
    auto stage_dot = [&](const ColRegs<TS_EPT2> &cur, int i) {
        const d2 *rs2 = reinterpret_cast<const d2 *>(r_s);
        double sd = 0.0;
#pragma unroll
        for (int k = 0; k < TS_EPT2; ++k) {
            const int e = k * TS_THREADS + tid;
            if (e < ld2) {
                const d2 rv = rs2[e];
                sd += cur.v[k].x * rv.x;
                sd += cur.v[k].y * rv.y;
            }
        }
        sd = wave_allreduce_sum(sd);
        double *slot = part + (i & 1) * TS_MAXWAVES;
        if (lane == 0) slot[wave] = sd;
        if (wave == 0 && lane < 6) scal[(i & (TS_RING - 1)) * 8 + lane] = cur.sc;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
#pragma unroll
            for (int w = 0; w < TS_WAVES; ++w) t += slot[w];
            st_gran(gteam + ((size_t)(i & (TS_RING - 1)) * TS_MAXQ + q) * 2, a.tag0 + (unsigned)i + 1u, t);
        }
    };

    // C(i): the Q parts of column i in member order -> gradient, leapfrog update, forward axpy.
    // false: a wait timed out (or another workgroup gave up): leave.
    auto stage_finish = [&](const ColRegs<TS_EPT2> &cur, int i) -> bool {
        if (wave == 0) {
            const unsigned tag = a.tag0 + (unsigned)i + 1u;
            const u64 *g = gteam + (size_t)(i & (TS_RING - 1)) * TS_MAXQ * 2;
            double t = 0.0;
            bool ok = true;
            unsigned spins = 0;
            long long t0 = 0;
            while (!poll_parts_scalar(g, a.poll_q, a.Q, tag, t)) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 63u) == 0) {
                    const long long now = wall_clock64();
                    if (t0 == 0) t0 = now;
                    if (__hip_atomic_load(a.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                        now - t0 > RES_TIMEOUT_TICKS) {
                        __hip_atomic_store(a.abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = false;
                        break;
                    }
                }
            }
            if (lane == 0) {
                tot_s[0] = t;
                if (!ok) *abort_s = 1;
            }
        }
        __syncthreads();
        if (*abort_s) return false;
        const int64_t j = jb + i;
        const double *sc = scal + (i & (TS_RING - 1)) * 8;
        const double cx = sc[0], cp = sc[1], clo = sc[2], chi = sc[3], cgr = sc[4], cpn = sc[5];
        const double g = 2.0 * tot_s[0] + cgr;
        const bool writer = (q == 0 && tid == 0);
        double xj = cx;
        if ((mode & SW_GOUT) && writer) s.g_out[j] = g;
        if (mode & SW_PFIN) {
            const double pf = cp - s.c_p * g;
            pp += pf * pf;
            if (!(mode & SW_SPEC) && writer) s.p_out[j] = pf;
        }
        if (mode & SW_UPD) {
            const double psrc = (mode & SW_SPEC) ? cpn : cp;
            double pj = psrc - s.c_u * g;
            xj = cx + s.dt * pj;
            if (xj > chi) {
                xj = chi;
                pj = -pj;
            } else if (xj < clo) {
                xj = clo;
                pj = -pj;
            }
            if (writer) {
                s.p_out[j] = pj;
                s.x_out[j] = xj;
            }
        }
        if (mode & SW_FWD) {
#pragma unroll
            for (int k = 0; k < TS_EPT2; ++k) {
                dacc[k].x += cur.v[k].x * xj;
                dacc[k].y += cur.v[k].y * xj;
            }
        }
        return true;
    };

    // TS_D column buffers rotate: finishing (i-1) | dotted (i) | in flight (i+1 .. i+TS_D-2).
    // Iteration i: request column i+TS_D-2 into the registers column i-2 left in the last iteration;
    // dot and publish column i; THEN collect the parts of column i-1 -- published a whole column
    // (~3 us of streaming) earlier, so the poll finds them at its first look -- and finish it.
    // (Collecting right after publishing instead exposes the exchange latency every column:
    // measured 18.6 against 17.0 ms per sweep of 96 GB.)
    ColRegs<TS_EPT2> B[TS_D];
    bool ok = true;
#pragma unroll
    for (int c0 = 0; c0 < TS_D - 2; ++c0)
        if (c0 < cnt) load_col(B[c0], c0);
    int i = 0;
    while (ok && i < cnt) {
#pragma unroll
        for (int r = 0; r < TS_D; ++r) {  // r == i % TS_D: every buffer index below is a constant
            if (i >= cnt) break;
            if (i + TS_D - 2 < cnt) load_col(B[(r + TS_D - 2) % TS_D], i + TS_D - 2);
            stage_dot(B[r], i);
            if (i > 0) {
                if (!(ok = stage_finish(B[(r + TS_D - 1) % TS_D], i - 1))) break;
            }
            ++i;
        }
    }
    if (ok && cnt > 0) {
        const int last = cnt - 1;
#pragma unroll
        for (int r = 0; r < TS_D; ++r)
            if (last % TS_D == r) ok = stage_finish(B[r], last);
    }
    if (!ok) return;

    if (mode & SW_PFIN) {
        if (q == 0 && tid == 0) s.pp_part[team] = pp;
        // (the host sums n_pp partials: the row-panel path's elementwise update writes more of them)
        if (blockIdx.x == 0)
            for (int t = 8 * a.tpx + tid; t < a.n_pp; t += TS_THREADS) s.pp_part[t] = 0.0;
    }
    if (mode & SW_FWD) {
        d2 *out = reinterpret_cast<d2 *>(s.slab + (int64_t)team * ld + row0);
#pragma unroll
        for (int k = 0; k < TS_EPT2; ++k) {
            const int e = k * TS_THREADS + tid;
            if (e < ld2) out[e] = dacc[k];
        }
    }
}

// placement probe: XCC_ID of every block of a grid shaped like the team sweep's
__global__ void __launch_bounds__(1024) team_probe_kernel(unsigned *xcc)
{
    if (threadIdx.x == 0) xcc[blockIdx.x] = (unsigned)__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xfu;  // HW_REG_XCC_ID[3:0]
}

}  // namespace ghk
