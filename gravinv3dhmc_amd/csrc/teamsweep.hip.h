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
// travels as two 8-byte granules {tag, 32 bits}, each written by one write-through store and read
// with sc1 loads; no counters, no fences; correct under any placement of the workgroups.  What
// keeps the stream of G going across the hand-off of every column:
//   * the next column is REALLY in flight while this one is processed.  The compiler counts
//     outstanding vector loads per path and waits for the fewest any path may have issued, so one
//     conditional vector load behind a column request (a guarded element, the per-column scalars
//     of lanes 0..5, a request under `if (i + 1 < cnt)`) turns every wait for a column into
//     s_waitcnt vmcnt(0): nothing is prefetched, each column pays the full memory latency plus
//     the hand-off (5.9 TB/s at the C5 share, where sweep_kernel -- same defect, shorter hand-off
//     -- still reached the plain-read rate).  Hence: unconditional column loads (threads past the
//     end re-read the last double2; past the last column the last column again), 32-bit offsets
//     from a uniform base held in their own registers, the scalars through the scalar cache
//     (s_load, lgkmcnt), the poll address from a persistent offset register.  The waits are then
//     vmcnt(5): the column just requested stays in flight;
//   * a lag of two columns: a member publishes its part of column i and then finishes column i-2.
//     The column dotted one iteration ago stays in its registers, the one before it waits in LDS
//     (every thread parks and fetches its own elements: no hazard between threads; 2 x 80 KB of
//     LDS traffic per column beside the HBM stream), at three register buffers.  With one column
//     of lag 17 % of the polls came before the part (each then pays a second poll that returns
//     behind the column requested meanwhile); with two, 1.5 %;
//   * the poll for column i-2 is ISSUED at the start of iteration i, in front of the request for
//     column i+1, and looked at after the dot of column i.  A wave's vector loads return in order:
//     issued there the poll comes back before column i+1's data; issued after the dot it would come
//     back behind it, a full column later.
// A ring of eight granule slots per member suffices (a member at column k has the parts of column
// k-2 of everybody: no member is more than two columns ahead of another, and the slowest still reads
// column k-4).  Members of a team are blocks with equal blockIdx % 8, i.e. on one XCD
// under the round-robin dispatch (their exchange then stays in that L2), but nothing depends on it.
// Every wait is bounded (2 s): on a time-out the abort word is raised, every workgroup leaves, later
// launches of the stream return at once and the host repeats the work in row panels
// (host_sweep.h: team_failed).
#pragma once

namespace ghk {

constexpr int TS_MAXQ = 8;
constexpr int TS_RING = 8;
constexpr int TS_MAXWAVES = 16;

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
// TS_THREADS * TS_EPT2 * 2), TS_D column buffers in registers (>= 3): finishing | dotted | TS_D - 2 in
// flight; TS_LAG2 (TS_D = 3): waiting | dotted | in flight, and one more waiting column in LDS.
template <int TS_THREADS, int TS_EPT2, int TS_D, bool TS_LAG2>
__global__ void __launch_bounds__(TS_THREADS) teamsweep_kernel(TeamArgs a)
{
    static_assert(!TS_LAG2 || TS_D == 3, "the two-column lag rotates three register buffers");
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
    // LDS: r of the member's rows | (TS_LAG2: the parked column) | dot partials, 2 x 16 | total, flag
    double *r_s = smem;
    d2 *park2 = reinterpret_cast<d2 *>(smem + a.panel_rows);
    double *part = smem + (TS_LAG2 ? 2 : 1) * a.panel_rows;
    double *tot_s = part + 2 * TS_MAXWAVES;
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

    // (unconditional column loads at 32-bit offsets, the scalars through the scalar cache: see
    // sweep_kernel -- a conditional vector load anywhere behind a column request turns every wait for
    // that column into vmcnt(0), and nothing is prefetched any more)
    unsigned coff[TS_EPT2];
#pragma unroll
    for (int k = 0; k < TS_EPT2; ++k) {
        const int e = k * TS_THREADS + tid;
        coff[k] = (unsigned)(e < ld2 ? e : (ld2 > 0 ? ld2 - 1 : 0)) * (unsigned)sizeof(d2);
    }
    const int64_t row0c = rows > 0 ? row0 : 0;
    const kconst_ptr kx = as_kconst(s.x_in), kp = as_kconst(s.p_in), klo = as_kconst(s.low), khi = as_kconst(s.high),
                     kgr = as_kconst(s.greg), kpn = as_kconst(s.pn_in);
    auto load_col = [&](ColRegs<TS_EPT2> &c, int i) {
        const char *col = reinterpret_cast<const char *>(s.G + (jb + i) * ld + row0c);
#pragma unroll
        for (int k = 0; k < TS_EPT2; ++k) {
            // (kept 32-bit next to the uniform base -- global_load v, voff, s[base] -- and in its own
            // register: as a copy it lands in the destination registers and waits for their last load)
            asm volatile("" : "+v"(coff[k]));
            const unsigned o = coff[k];
            c.v[k] = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(col + o));
        }
    };

    // the member's granule pair of column i (ring slot i & 3; the Q pairs of a slot are contiguous)
    auto gran_of = [&](int member, int i) -> u64 * {
        return gteam + ((size_t)(i & (TS_RING - 1)) * TS_MAXQ + member) * 2;
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
        sd = wave_sum_dpp(sd);  // on the VALU: the hand-off chain of a column starts here
        double *slot = part + (i & 1) * TS_MAXWAVES;
        if (lane == 0) slot[wave] = sd;
        __syncthreads();
        if (wave == 0) {
            // the waves' partials: one LDS read per lane, a row scan, lane 15 publishes
            const double t = row16_sum_dpp(lane < TS_WAVES ? slot[lane] : 0.0);
            if (lane == 15) st_gran(gran_of(q, i), a.tag0 + (unsigned)i + 1u, t);
        }
    };

    // issue the poll for column i (wave 0, lane = member): the two words come back with the loads
    // already in flight; they are looked at in stage_finish (looking at them here would hold the
    // wave's column request back until they have arrived)
    unsigned poff = (unsigned)lane * (unsigned)(2 * sizeof(u64));  // the lane's member in a ring slot
    auto poll_issue = [&](int i, u64 &pa, u64 &pb) {
        pa = pb = 0;
        if (wave == 0 && lane < a.poll_q) {
            // (uniform base + the lane's offset from its own register: an address built in a scratch
            // register lands in a column buffer's registers and waits for that buffer's loads)
            asm volatile("" : "+v"(poff));
            u64 *g = reinterpret_cast<u64 *>(reinterpret_cast<char *>(gran_of(0, i)) + poff);
            ld_gran_issue(g, pa, pb);
        }
        __builtin_amdgcn_sched_barrier(0);  // in front of the column request that follows
    };

    // C(i): the Q parts of column i in member order -> gradient, leapfrog update, forward axpy.
    // pa / pb: what the early poll brought.  false: a wait timed out (or another workgroup gave
    // up): leave.
    auto stage_finish = [&](const ColRegs<TS_EPT2> &cur, int i, u64 pa, u64 pb, bool parked) -> bool {
        // the column's scalars, through the scalar cache (every wave its own; back by the barrier below)
        const int64_t j = jb + i;
        const double cx = kx ? kx[j] : 0.0, cp = kp ? kp[j] : 0.0, clo = klo ? klo[j] : 0.0, chi = khi ? khi[j] : 0.0,
                     cgr = kgr ? kgr[j] : 0.0, cpn = kpn ? kpn[j] : 0.0;
        if (wave == 0) {
            bool ok = true;
            double pv = 0.0;
            const bool pok = lane >= a.poll_q || gran_value(pa, pb, a.tag0 + (unsigned)i + 1u, pv);
            if (!__all(pok)) {
                // (a member lags by more than the early poll allows for) poll until every part is there
                if (lane == 0) __hip_atomic_fetch_add(a.abort_w + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = res_poll(a.abort_w, [&]() {
                    return lane >= a.poll_q || ld_gran(gran_of(lane, i), a.tag0 + (unsigned)i + 1u, pv);
                });
            }
            double t = 0.0;
            for (int m = 0; m < a.Q; ++m) t += __shfl(pv, m, WAVE);
            if (lane == 0) {
                tot_s[0] = t;
                if (!ok) *abort_s = 1;
            }
        }
        __syncthreads();
        if (*abort_s) return false;
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
            if (TS_LAG2 && parked) {
                // the column comes out of the thread's own LDS slots
#pragma unroll
                for (int k = 0; k < TS_EPT2; ++k) {
                    const int e = k * TS_THREADS + tid;
                    if (e < ld2) {
                        const d2 v = park2[e];
                        dacc[k].x += v.x * xj;
                        dacc[k].y += v.y * xj;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < TS_EPT2; ++k) {
                    dacc[k].x += cur.v[k].x * xj;
                    dacc[k].y += cur.v[k].y * xj;
                }
            }
        }
        return true;
    };

    // the thread's elements of a column into its LDS slots (after it fetched the previous one's)
    auto park = [&](const ColRegs<TS_EPT2> &c) {
#pragma unroll
        for (int k = 0; k < TS_EPT2; ++k) {
            const int e = k * TS_THREADS + tid;
            if (e < ld2) park2[e] = c.v[k];
        }
    };

    // TS_D column buffers rotate: finishing (i-1) | dotted (i) | in flight (i+1 .. i+TS_D-2).
    // Iteration i: issue the poll for column i-LAG, request column i+TS_D-2 into the registers freed
    // in the last iteration, dot and publish column i, finish column i-LAG (TS_LAG2: from LDS, and
    // column i-1 takes its place there).
    constexpr int LAG = TS_LAG2 ? 2 : 1;
    ColRegs<TS_EPT2> B[TS_D];
    bool ok = true;
    // (every iteration requests a column -- past the end the last one again: a request under a
    // condition would turn the waits for the columns in flight into vmcnt(0))
    const int lastc = cnt - 1;
#pragma unroll
    for (int c0 = 0; c0 < TS_D - 2; ++c0)
        if (cnt > 0) load_col(B[c0], c0 < lastc ? c0 : lastc);
    int i = 0;
    while (ok && i < cnt) {
#pragma unroll
        for (int r = 0; r < TS_D; ++r) {  // r == i % TS_D: every buffer index below is a constant
            if (i >= cnt) break;
            u64 pa = 0, pb = 0;
            if (i >= LAG) poll_issue(i - LAG, pa, pb);
            load_col(B[(r + TS_D - 2) % TS_D], i + TS_D - 2 < lastc ? i + TS_D - 2 : lastc);
            stage_dot(B[r], i);
            if (i >= LAG) {
                if (!(ok = stage_finish(B[(r + TS_D - 1) % TS_D], i - LAG, pa, pb, TS_LAG2))) break;
            }
            if (TS_LAG2 && i >= 1) park(B[(r + TS_D - 1) % TS_D]);
            ++i;
        }
    }
    if (ok && TS_LAG2 && cnt > 1) {
        u64 pa = 0, pb = 0;
        poll_issue(cnt - 2, pa, pb);
        ok = stage_finish(B[0], cnt - 2, pa, pb, true);
    }
    if (ok && cnt > 0) {
        const int last = cnt - 1;
        u64 pa = 0, pb = 0;
        poll_issue(last, pa, pb);
#pragma unroll
        for (int r = 0; r < TS_D; ++r)
            if (last % TS_D == r) ok = stage_finish(B[r], last, pa, pb, false);
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

}  // namespace ghk
