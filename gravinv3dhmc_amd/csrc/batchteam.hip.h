// Several HMC chains on the STORED kernel, one read of G per leapfrog step: teams of workgroups (gfx950).
//
// The two-pass batch (batch.hip.h) reads G twice per step -- adjoint of all chains, update, forward --
// because the update between the two products needs the dots over ALL rows and a 16-column tile of G
// (C2: 1.3 MB) cannot wait on one CU.  It can wait on a TEAM: the workgroups that hold the row chunks
// of the same column tiles (grid = members x ranges of column tiles, every workgroup resident, one per
// CU) read their 448 rows x 16 columns of a tile ONCE into registers in MFMA operand order, contract
// them with their rows of the residuals (kept in registers for the whole launch), hand the team their
// 16 x 16 partial dots, and two tiles later multiply the same values -- parked in LDS meanwhile -- by
// the updated positions.  Both products of all chains from one read of G (hmc.py:114-152 for every
// chain; potential.py:698,708 are the two products).
//
// Exchange (the data is the flag, resident.hip.h: a double travels as two tagged 8-byte granules,
// written through, read with agent-scope loads, no fences, correct under any placement), a
// reduce-scatter and an all-gather of one hop each:
//   iteration it      every member publishes its 256 partial dots of tile it
//   iteration it + 1  member m sums the team's parts of ITS nval = ceil(256 / members) (cell, chain)
//                     pairs (lane = member, row scans: a fixed order), applies their leapfrog updates,
//                     stores the new state and publishes the new positions
//   iteration it + 2  every member collects the 256 new positions and runs the forward MFMAs on the
//                     tile it parked in LDS two iterations ago
// A member reads 2 x 4 KB per tile from the team, against 57 KB of G (all members collecting all parts
// would read 92 KB).  Rings of four slots; every wait is bounded (2 s): on a time-out the abort word is
// raised, every workgroup leaves, and the host repeats the work with the two-pass kernels
// (mfb_fused_failed).
//
// Loads return in order (one vmcnt counter): waiting for a small load of the exchange also waits for
// every load of G issued before it.  So per iteration the small loads (the team's parts, the new
// positions, the operands of the updates) are issued FIRST and waited for in ONE place, and the tile
// two iterations ahead is requested right behind that wait: it has a whole iteration to arrive before
// the next wait, and three register sets of tiles rotate (the loop is unrolled by three).  Eight waves
// of 256 registers rather than sixteen of 128: the tiles, the residuals and the forward accumulators
// are 160 registers of a lane; what a wave needs besides is paid once, not twice.
//
// Per iteration and CU: 57 KB of G from HBM, 112 + 112 v_mfma_f64_16x16x4 (1.5 us of the four matrix
// pipes at 64 cycles each), three barriers.  At 16 chains the step is within 20 % of BOTH roofs of the
// chip (40 GB at 8 TB/s = 5.0 ms; 3.2e11 flop at 78.6 TFLOP/s = 4.1 ms).
#pragma once

namespace ghk {

constexpr int BT_RC = 7;            // row blocks of 64 per member: 448 rows, 28 patches of 16
constexpr int BT_NW = 8;            // waves of a workgroup
constexpr int BT_PPW = (MfbTile<BT_RC>::PATCHES + BT_NW - 1) / BT_NW;  // row patches per wave: 4 (3 for waves 4 .. 7)
static_assert(BT_PPW == 4, "bt_wait_but: six or eight requests per wave");
static_assert(BT_NW * (BT_PPW - 1) < MfbTile<BT_RC>::PATCHES, "every wave has its first BT_PPW - 1 patches");
constexpr int BT_MINMEM = 8;        // a member updates ceil(256 / members) <= 32 pairs of a tile
constexpr int BT_MAXMEM = 32;       // N <= 14336 rows
constexpr int BT_RING = 4;
// column stride of a parked tile (doubles): 466 * 8 = 144 (mod 256) bytes -- the 16 columns of a 16-byte park
// write fall on 16 different bank groups, the four column groups of the forward's 8-byte reads overlap in few
constexpr int BT_S = MfbTile<BT_RC>::ROWS + 18;
constexpr int BT_BUF = 16 * BT_S;
constexpr size_t BT_LDS = (2 * (size_t)BT_BUF + BT_NW * 256 + 256) * sizeof(double);  // 137728 B

struct BtArgs {
    int tiles_per_range;
    int nval;            // (cell, chain) pairs of a tile a member owns: ceil(256 / members)
    double *slab;        // gridDim.y x (ld x 16): forward partials per range
    u64 *gran_p;         // [gridDim.y][BT_RING][BT_MAXMEM][256][2]: the members' partial dots
    u64 *gran_x;         // [gridDim.y][BT_RING][256][2]: the new positions
    unsigned tag0;       // tags tag0 + 1 .. tag0 + tiles_per_range belong to this launch
    unsigned *abort_w;
    int poll_members;    // gridDim.x (+ 1 in the time-out test: one part never comes)
    int n_pp;            // rows of pp_part the host sums
    long long *dbg;      // optional: accumulated phase times (100 MHz ticks) of lane 0 of wave dbg_wave of member dbg_mem, range 0
    int dbg_mem, dbg_wave;
    int dbg_break;       // timing experiments that BREAK results (GRAVHMC_BT_BREAK): 1 no requests for G inside the loop, 2 forward MFMAs on constants instead of LDS operands, 4 no parking
};

template <int V> struct BtIC { static constexpr int value = V; };

// Workgroup barrier that orders LDS only.  __syncthreads() carries a release fence: a wait for EVERY memory
// operation of the wave in flight -- the tile of G requested a moment ago (a whole HBM round trip), the
// acknowledgements of the write-through stores of the exchange.  What the waves of a workgroup hand each
// other across these barriers lives in LDS.
// The loads of this kernel as assembly, invisible to the compiler's wait-count bookkeeping: loads return in
// order, and a wait the compiler sets for a small load of the exchange would count the requests for G issued
// behind it as if they were in front -- a whole HBM round trip.  The waits that cover them are explicit
// s_waitcnt vmcnt(n) with n = the number of requests for G issued since (bt_wait_*).  No instruction may touch
// a destination register between such a load and its wait: tests/test_host.py checks the generated code.
__device__ __forceinline__ void bt_ld_gran_asm(const u64 *g, u64 &a, u64 &b)
{
    asm volatile("global_load_dwordx2 %0, %1, off sc1" : "+v"(a) : "v"(g));
    asm volatile("global_load_dwordx2 %0, %1, off offset:8 sc1" : "+v"(b) : "v"(g));
}
__device__ __forceinline__ void bt_wait_all() { __builtin_amdgcn_s_waitcnt(0x0f70); }        // vmcnt(0)
__device__ __forceinline__ void bt_wait_but(int n)                                           // vmcnt(n), n = 6 | 8
{
    if (n == 8)
        __builtin_amdgcn_s_waitcnt(0x0f78);
    else
        __builtin_amdgcn_s_waitcnt(0x0f76);
}

__device__ __forceinline__ void bt_lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__global__ void __launch_bounds__(BT_NW * 64) batch_team_kernel(BatchAdjArgs a, BtArgs f)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ double ppred[32];
    __shared__ double chs[3][16];  // phase, cu, cp of the chains
    __shared__ long long tph_s[10];
    __shared__ int abort_s;
    using TL = MfbTile<BT_RC>;
    constexpr int NT = BT_NW * 64;
    double *red = smem + 2 * BT_BUF;    // BT_NW x 256: the waves' adjoint accumulators
    double *xs_s = red + BT_NW * 256;   // 256: new positions of the tile being finished, [col][chain]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lo = lane & 15, k = lane >> 4;
    const int mem = blockIdx.x, cr = blockIdx.y, members = (int)gridDim.x;
    const int64_t ntiles = (a.M + 15) / 16;
    const int64_t t0 = (int64_t)cr * f.tiles_per_range;
    const int ntl = (int)((ntiles - t0 < f.tiles_per_range) ? ntiles - t0 : f.tiles_per_range);
    if (tid == 0) abort_s = (__hip_atomic_load(f.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) ? 1 : 0;
    if (tid < 16) {
        chs[0][tid] = (double)a.phase[tid];
        chs[1][tid] = a.cu[tid];
        chs[2][tid] = a.cp[tid];
    }
    if (tid < 32) ppred[tid] = 0.0;
    // row patches of this wave: wave + 8 i (seven per SIMD); patches past the member's / the matrix's last one
    // re-read patch 0 of the matrix against zero residuals and are never stored
    bool pin[BT_PPW], pok[BT_PPW];
    int gpa[BT_PPW];
    const d2 *rt = reinterpret_cast<const d2 *>(a.Rt) + (k * 16 + lo);
    const d2 zero2 = d2{0.0, 0.0};
    // the member's rows of the residuals, MFMA operand order (batch.hip.h), for the whole launch
    d2 rr[BT_PPW][2];
    int64_t goff[BT_PPW];
#pragma unroll
    for (int i = 0; i < BT_PPW; ++i) {
        const int p = wave + BT_NW * i;
        const int gp = mem * TL::PATCHES + p;
        pin[i] = i < BT_PPW - 1 || p < TL::PATCHES;  // (uniform: the wave has an i-th patch -- only the last one may be missing)
        pok[i] = pin[i] && gp < a.np;
        gpa[i] = pok[i] ? gp : 0;
        rr[i][0] = rt[128 * gpa[i]];
        rr[i][1] = rt[128 * gpa[i] + 64];
        if (!pok[i]) rr[i][0] = rr[i][1] = zero2;
        // lane (lo, k) feeds column lo, rows 2 k, 2 k + 1 and 8 + 2 k, 8 + 2 k + 1 of a patch: four lanes cover
        // 64 contiguous bytes of a column per load, a member's rows of a column are one run of 3.5 KB
        goff[i] = (int64_t)16 * gpa[i] + 2 * k;
    }
    d2 ts[3][BT_PPW][2];
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int i = 0; i < BT_PPW; ++i) ts[q][i][0] = ts[q][i][1] = zero2;
    auto tile_col = [&](int64_t tile) -> const double * {
        int64_t j = tile * 16 + lo;
        if (j >= a.M) j = a.M - 1;
        if (j < 0) j = 0;
        return a.G + j * a.ld;
    };
    auto tile_load_piece = [&](d2 (&t)[BT_PPW][2], const double *col, int i, int h) {
        const double *ptr = col + goff[i];
        if (h == 0)
            asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(t[i][0]) : "v"(ptr));
        else
            asm volatile("global_load_dwordx4 %0, %1, off offset:64 nt" : "+v"(t[i][1]) : "v"(ptr));
    };
    auto tile_load = [&](d2 (&t)[BT_PPW][2], int64_t tile) {
        int64_t j = tile * 16 + lo;
        if (j >= a.M) j = a.M - 1;
        const double *col = a.G + j * a.ld;
        // (as assembly: the compiler's wait-count bookkeeping loses track of these loads across the rotating
        // sets and the polling loops and puts vmcnt(0) in front of the MFMAs -- behind the small loads just
        // issued.  The waits that cover them are the explicit ones: before the loop and in every exchange
        // phase, one iteration after the request and one before the use.)
#pragma unroll
        for (int i = 0; i < BT_PPW; ++i)
            if (pin[i]) {
                const double *ptr = col + goff[i];
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "+v"(t[i][0]) : "v"(ptr));
                asm volatile("global_load_dwordx4 %0, %1, off offset:64 nt" : "+v"(t[i][1]) : "v"(ptr));
            }
    };
    u64 *gpt = f.gran_p + (size_t)cr * BT_RING * BT_MAXMEM * 512;
    u64 *gxt = f.gran_x + (size_t)cr * BT_RING * 512;
    auto gran_p_of = [&](int it, int member) -> u64 * {
        return gpt + ((size_t)(it & (BT_RING - 1)) * BT_MAXMEM + member) * 512;
    };
    auto gran_x_of = [&](int it) -> u64 * { return gxt + (size_t)(it & (BT_RING - 1)) * 512; };
    // Reduce-scatter roles: wave 7 - gw sums this member's pairs, lane = member of the team: teams of more
    // than 16 members two pairs per wave (half a wave each), smaller ones four (a row of 16 lanes each).
    const bool wide = members > 16;
    const int gw = BT_NW - 1 - wave;
    const int gpw = wide ? 2 : 4;                          // pairs per wave
    const bool gwave = gw * gpw < f.nval;                  // (uniform)
    const int gm = wide ? (lane & 31) : (lane & 15);
    const int gslot = wide ? 2 * gw + (lane >> 5) : 4 * gw + (lane >> 4);
    const int gv = mem * f.nval + gslot;                   // the pair: column gv >> 4 of the tile, chain gv & 15
    const bool gact = gslot < f.nval && gv < 256;
    const int gvc = gact ? gv : 0;
    const int gmc = gm < f.poll_members ? gm : 0;
    const bool glive = a.phase[gvc & 15] != PH_IDLE;       // (idle chains -- and the unused slots of a batch of < 16 -- exchange nothing)
    const bool gask = gact && glive && gm < f.poll_members;
    const bool ulane = gact && gm == (wide ? 31 : 15);
    const bool lead = wave < TL::PATCHES - BT_NW * (BT_PPW - 1);   // waves 0 .. 3: four patches, the SIMD's first wave
    const bool xwave = wave < 4;                           // threads 0 .. 255: one per pair of a tile
    const bool xlive = a.phase[tid & 15] != PH_IDLE;
    // the arrays the updates store to, as wave-uniform values fetched ONCE (chosen by a lane's phase they become
    // a load from the kernel-argument segment per update -- a memory round trip behind the tile just requested)
    auto uniform_ptr = [](double *p) -> double * {
        const unsigned long long v = (unsigned long long)p;
        const unsigned lo32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
        const unsigned hi32 = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
        return (double *)(((unsigned long long)hi32 << 32) | lo32);
    };
    // (...and kept in vector registers behind an opaque barrier: as scalar values the compiler re-reads them from
    // the kernel-argument segment inside the loop when scalar registers run short -- a memory round trip in front
    // of every update)
    double *Xo = uniform_ptr(a.X_out), *Po = uniform_ptr(a.P_out), *Go = uniform_ptr(a.G_out);
    double dt = a.dt;
    unsigned tag0 = f.tag0;
    const double *greg_p = a.GREG, *pn_p = a.Pn;
    asm volatile("" : "+v"(Xo), "+v"(Po), "+v"(Go), "+v"(dt), "+v"(tag0));
    const bool has_greg = greg_p != nullptr, has_pn = pn_p != nullptr;
    d4 accf[BT_PPW];
#pragma unroll
    for (int i = 0; i < BT_PPW; ++i) accf[i] = d4{0.0, 0.0, 0.0, 0.0};
    double pp = 0.0;
    __syncthreads();
    if (abort_s) return;
    const bool clk = f.dbg != nullptr && mem == f.dbg_mem && cr == 0 && tid == 64 * f.dbg_wave;
    if (clk) {
        for (int q = 0; q < 8; ++q) tph_s[q] = 0;
        tph_s[9] = wall_clock64();
    }
    auto mark = [&](int ph) {
        if (clk) {
            const long long now = wall_clock64();
            tph_s[ph] += now - tph_s[9];
            tph_s[9] = now;
        }
    };
    // bounded wait for one double of the exchange (the loads of the first look were issued a phase ago)
    auto settle = [&](u64 *g, u64 ga, u64 gb, unsigned tag, double &val) -> bool {
        if (gran_value(ga, gb, tag, val)) return true;
        unsigned spins = 0;
        long long tstart = 0;
        while (!ld_gran(g, tag, val)) {
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 63u) == 0) {
                const long long now = wall_clock64();
                if (tstart == 0) tstart = now;
                if (__hip_atomic_load(f.abort_w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u ||
                    now - tstart > RES_TIMEOUT_TICKS) {
                    __hip_atomic_store(f.abort_w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    return false;
                }
            }
        }
        return true;
    };
    if (ntl > 0) tile_load(ts[0], t0);
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): tile 0 is there
    if (ntl > 1) tile_load(ts[1], t0 + 1);
    // The small loads of iteration itn, issued at its start (what they fetch was published before the forward
    // MFMAs of the iteration before: a round trip earlier) and looked at behind its adjoint MFMAs: the team's parts of
    // tile itn - 1 and the operands of this member's updates of it; the new positions of tile itn - 2.  Every
    // one unconditional inside a uniform branch (a load under a lane condition costs a wait for everything
    // in flight where its register is preset).
    u64 ga = 0, gb = 0, xa = 0, xb = 0;
    double unv = 0.0;  // lane q < 6 of a pair's lanes: the q-th operand of the pair's update (x, p, greg, pn, high, low)
    // (what a lane's addresses do not share with the iteration, formed once: pointers taken from the kernel
    // arguments inside the loop are scalar loads from the argument segment -- a memory round trip each)
    const u64 *const gp_lane = gpt + (size_t)gmc * 512 + 2 * gvc;   // + slot * BT_MAXMEM * 512
    const u64 *const gx_lane = gxt + 2 * tid;                        // + slot * 512
    // the operand this lane fetches for its pair: base (at cell 0), stride per cell
    const double *un_base;
    int un_cell_stride;
    {
        const int c16 = gvc & 15;
        const double *xin = a.X_in + c16, *pin_ = a.P_in + c16;
        un_base = gm == 0 ? xin : gm == 1 ? pin_ : gm == 2 ? (a.GREG ? a.GREG + c16 : xin)
                : gm == 3 ? (a.Pn ? a.Pn + c16 : xin) : gm == 4 ? a.high : a.low;
        un_cell_stride = gm < 4 ? CB : 1;
    }
    const int64_t Mcells = a.M;
    auto request = [&](int itn) {
        const bool f1n = itn >= 1 && itn - 1 < ntl, f2n = itn >= 2 && itn <= ntl + 1;
        if (f1n && gwave) {
            const int64_t uj = (t0 + itn - 1) * 16 + (gvc >> 4);   // the pair's cell
            const int64_t ujc = uj < Mcells ? uj : 0;
            bt_ld_gran_asm(gp_lane + (size_t)((itn - 1) & (BT_RING - 1)) * (BT_MAXMEM * 512), ga, gb);
            const double *src = un_base + ujc * un_cell_stride;
            asm volatile("global_load_dwordx2 %0, %1, off" : "+v"(unv) : "v"(src));
        }
        if (f2n && xwave) bt_ld_gran_asm(gx_lane + (size_t)((itn - 2) & (BT_RING - 1)) * 512, xa, xb);
    };
    // One iteration; SC: the register set of tile it (it % 3).  false: the team gave up.
    //   small requests | park tile it - 1 in LDS | adjoint MFMAs of tile it | exchange: new positions of tile
    //   it - 2 to LDS, sums, updates and new positions of tile it - 1; request tile it + 2 | barrier | publish
    //   the parts of tile it | forward MFMAs of tile it - 2 | barrier
    // The matrix pipes are the long pole (2 x 28 MFMAs per SIMD and iteration); LDS traffic and memory
    // round trips sit next to MFMAs of the same or the SIMD's other wave.
    auto body = [&](auto SC, int it) -> bool {
        constexpr int S0 = decltype(SC)::value, SP = (S0 + 2) % 3;
        const bool stg = it < ntl, f1 = it >= 1 && it - 1 < ntl, f2 = it >= 2;
        request(it);

        if (f1 && !(f.dbg_break & 4)) {
            double *buf = smem + (size_t)((it - 1) & 1) * BT_BUF + lo * BT_S + 2 * k;
#pragma unroll
            for (int i = 0; i < BT_PPW; ++i)
                if (pin[i]) {
                    *reinterpret_cast<d2 *>(buf + 16 * (wave + BT_NW * i)) = ts[SP][i][0];
                    *reinterpret_cast<d2 *>(buf + 16 * (wave + BT_NW * i) + 8) = ts[SP][i][1];
                }
        }
        mark(0);
        // ---- adjoint MFMAs on the registers of tile it: this member's part of S (one accumulator per patch:
        // consecutive MFMAs are independent)
        const bool tl = it + 2 < ntl && !(f.dbg_break & 1);
        if (stg) {
            // (two accumulators: consecutive MFMAs are independent.)  The first wave of every SIMD (four patches):
            // behind every second MFMA one request for the tile two iterations ahead, into the set just parked.
            // The CU's address unit needs ~1 us for the 57 KB of requests of a tile, and a wave whose request
            // waits for it issues no MFMA either: the SIMD's other wave (three patches) keeps the matrix pipe
            // busy meanwhile and places ITS requests between its forward MFMAs, where the first wave is free.
            const double *tcol = tile_col(t0 + it + 2);
            d4 acc0 = d4{0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int xy = 0; xy < 2; ++xy)
#pragma unroll
                    for (int i = 0; i < BT_PPW; ++i) {
                        if (pin[i]) {
                            d4 &acc = (i & 1) ? acc1 : acc0;
                            acc = mfma_f64(xy ? ts[S0][i][h].y : ts[S0][i][h].x, xy ? rr[i][h].y : rr[i][h].x, acc);
                        }
                        const int n = (h * 2 + xy) * BT_PPW + i;   // 0 .. 15
                        if (n & 1) {
                            const int piece = n >> 1;                  // 0 .. 7: patch piece >> 1, half piece & 1
                            if (tl && lead) tile_load_piece(ts[SP], tcol, piece >> 1, piece & 1);
                        }
                        __builtin_amdgcn_sched_barrier(0);  // (keep the order: the scheduler strings one accumulator's MFMAs together)
                    }
            mark(1);
#pragma unroll
            for (int q = 0; q < 4; ++q) red[wave * 256 + q * 64 + lane] = acc0[q] + acc1[q];
        }

        // ---- exchange.  Everything but this iteration's requests for G has arrived -- said explicitly, before
        // anything is looked at or stored (a store's acknowledgement counts like a load): the tile requested an
        // iteration ago, which the next iteration multiplies, and the small loads issued at the start.
        if (tl && lead)
            bt_wait_but(2 * BT_PPW);   // (the eight requests of a moment ago)
        else
            bt_wait_all();
        bool ok = true;
        if (f2 && xwave) {
            double val = 0.0;
            ok = settle(gran_x_of(it - 2) + 2 * tid, xa, xb, tag0 + (unsigned)it - 1u, val);
            xs_s[tid] = val;
        }
        if (!ok) abort_s = 1;
        mark(2);
        bt_lds_barrier();
        mark(3);
        if (abort_s) return false;
        if (stg && xwave && xlive) {
            // pair tid = (column tid >> 4, chain tid & 15); acc[q] of lane (lo, k) is column k + 4 q, chain lo
            const int cl = tid >> 4, c = tid & 15;
            const int ridx = (cl >> 2) * 64 + (cl & 3) * 16 + c;
            double sp = 0.0;
#pragma unroll
            for (int w = 0; w < BT_NW; ++w) sp += red[w * 256 + ridx];
            st_gran(gran_p_of(it, mem) + 2 * tid, tag0 + (unsigned)it + 1u, sp);
        }
        mark(4);
        // ---- this member's pairs of tile it - 1: sums over the team, updates, new positions.  BEHIND the barrier:
        // what the barrier waits for is the waves' accumulators and the positions in LDS -- with the sums in front
        // of it, a round of the team (publish, become visible, load, update) sets the pace of the iterations.
        bool ok2 = true;
        double val = 0.0;
        double uo[6];
        if (f1 && gwave) {
            if (gask) ok2 = settle(gran_p_of(it - 1, gmc) + 2 * gvc, ga, gb, tag0 + (unsigned)it, val);
            const int pair0 = lane & (wide ? 32 : 48);   // first lane of this lane's pair
#pragma unroll
            for (int q = 0; q < 6; ++q) uo[q] = __shfl(unv, pair0 + q, WAVE);
        }
        if (f1 && gwave) {
            double tot = row16_sum_dpp(val);
            if (wide) tot = dpp_add<0x142, 0xa>(tot);  // lanes 31, 63: the 32 lanes' sum
            if (ulane) {
                const int c = gv & 15;
                const int64_t uj = (t0 + it - 1) * 16 + (gv >> 4);
                double xn = 0.0;
                if (uj < a.M) {
                    // the pair's leapfrog update by its chain's phase (mfb_update_pre's arithmetic; hmc.py:114-152)
                    const int64_t idx = uj * CB + c;
                    const int ph = (int)chs[0][c];
                    const double cu = chs[1][c], cp = chs[2][c];
                    const double gr = 2.0 * tot + (has_greg ? uo[2] : 0.0);
                    double xo = uo[0], po = uo[1];
                    if (ph == PH_GOUT) {
                        Go[idx] = gr;
                    } else {
                        if (ph == PH_UPD || ph == PH_PFIN_SPEC) {
                            if (ph == PH_PFIN_SPEC) {
                                const double pf = po - cp * gr;
                                pp += pf * pf;
                                po = has_pn ? uo[3] : 0.0;
                            }
                            double pj = po - cu * gr;
                            double xj = xo + dt * pj;
                            if (xj > uo[4]) {
                                xj = uo[4];
                                pj = -pj;
                            } else if (xj < uo[5]) {
                                xj = uo[5];
                                pj = -pj;
                            }
                            po = pj;
                            xo = xj;
                        } else if (ph == PH_PFIN) {
                            const double pf = po - cp * gr;
                            pp += pf * pf;
                            po = pf;
                        }
                        Po[idx] = po;
                        Xo[idx] = xo;
                    }
                    xn = xo;
                }
                st_gran(gran_x_of(it - 1) + 2 * gv, tag0 + (unsigned)it, xn);
            }
        }
        if (!ok2) abort_s = 1;   // (seen behind the next barrier)
        mark(5);
        // ---- forward MFMAs on the tile parked an iteration ago; the second wave of every SIMD requests its six
        // pieces of the tile two iterations ahead between them (they have the rest of this iteration and the
        // next one's adjoint MFMAs to arrive: the next exchange waits for everything)
        {
            const double *pbuf = smem + (size_t)(it & 1) * BT_BUF;
            const double *tcol = tile_col(t0 + it + 2);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                // (iterations 0 and 1 have no tile to multiply: the same code with zero positions -- the requests
                // must not sit in two places, or the compiler joins their registers with copies of data in flight)
                const double b = f2 ? xs_s[(4 * u + k) * 16 + lo] : 0.0;
                const double *sa = pbuf + (4 * u + k) * BT_S + lo;
#pragma unroll
                for (int i = 0; i < BT_PPW; ++i) {
                    if (pin[i]) accf[i] = mfma_f64((f2 && !(f.dbg_break & 2)) ? sa[16 * (wave + BT_NW * i)] : 1.0, b, accf[i]);
                    const int n = u * BT_PPW + i;
                    if ((n & 1) && (n >> 1) < 2 * (BT_PPW - 1)) {
                        const int piece = n >> 1;                  // 0 .. 5
                        if (tl && !lead) tile_load_piece(ts[SP], tcol, piece >> 1, piece & 1);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        mark(6);
        bt_lds_barrier();  // (tile it - 1 is parked into the buffer this forward read ... two iterations from now; red, xs_s free)
        mark(7);
        return true;
    };
    bool alive = true;
    for (int it = 0; alive && it <= ntl + 1; it += 3) {
        alive = body(BtIC<0>{}, it);
        if (alive && it + 1 <= ntl + 1) alive = body(BtIC<1>{}, it + 1);
        if (alive && it + 2 <= ntl + 1) alive = body(BtIC<2>{}, it + 2);
    }
    if (!alive) return;
    if (clk)
        for (int q = 0; q < 8; ++q) f.dbg[q] += tph_s[q];
    // forward partials of this member's rows: acc[q] of lane (lo, k) is row k + 4 q of the patch, chain lo
    double *out = f.slab + (int64_t)cr * a.ld * CB;
#pragma unroll
    for (int i = 0; i < BT_PPW; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int64_t row = (int64_t)16 * (mem * TL::PATCHES + wave + BT_NW * i) + k + 4 * q;
            if (pok[i] && row < a.ld) out[row * CB + lo] = accf[i][q];
        }
    // sum of p^2 after the final half step: this member's pairs by chain, in slot order
    if (a.pp_part) {
        if (ulane) ppred[gslot] = pp;
        __syncthreads();
        const int wg = cr * members + mem;
        if (tid < 16 && (a.phase[tid] == PH_PFIN || a.phase[tid] == PH_PFIN_SPEC)) {
            double t = 0.0;
            for (int s = 0; s < f.nval && s < 32; ++s) {
                const int v = mem * f.nval + s;
                if (v < 256 && (v & 15) == tid) t += ppred[s];
            }
            a.pp_part[(int64_t)wg * CB + tid] = t;
        }
        // (the host sums n_pp rows: the two-pass adjoint writes more of them)
        if (wg == 0) {
            const int first = members * (int)gridDim.y;
            for (int e = first * CB + tid; e < f.n_pp * CB; e += NT) {
                const int ph = a.phase[e & 15];
                if (ph == PH_PFIN || ph == PH_PFIN_SPEC) a.pp_part[e] = 0.0;
            }
        }
    }
}

}  // namespace ghk
