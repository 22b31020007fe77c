// libgravhmc host side: buffers and evaluation of the fp64-MFMA chain batch (batch.hip.h).
// Included once by gravhmc.hip.
#pragma once

// --------------------------------------------------------------- batched chains (MFMA)

// ------------------------------------------------ batched chains on the matrix-free kernel (mfbatch.hip.h)

static int mfb_kind(const gh_ctx *c)
{
    if (c->cell_kind != GH_CELL_TESSEROID) return 0;
    if (!c->mf_near_on) return 1;
    if (c->mf_exact) return 2;
    return (c->obs_h_uniform && env_int("GRAVHMC_MFB_RU", 1) != 0) ? 4 : 3;  // 4: one observation height
}

typedef void (*mfb_adj_fn)(MfGeom, BatchAdjArgs, const double *, const double *, const double *, MfStats *);
typedef void (*mfb_fwd_fn)(MfGeom, MfbFwdArgs, const double *, MfStats *);

static mfb_adj_fn mfb_adj_for(const gh_ctx *c)
{
    switch (mfb_kind(c)) {
    case 0: return mfb_adjoint_kernel<0>;
    case 1: return mfb_adjoint_kernel<1>;
    case 2: return mfb_adjoint_kernel<2>;
    case 3: return mfb_adjoint_kernel<3>;
    default: return mfb_adjoint_kernel<4>;
    }
}

typedef void (*mfb_fus_fn)(MfGeom, BatchAdjArgs, MfbFusArgs, const double *, const double *, const double *, MfStats *);

static mfb_fus_fn mfb_fus_for(const gh_ctx *c)
{
    switch (mfb_kind(c)) {
    case 0: return mfb_fused_kernel<0>;
    case 1: return mfb_fused_kernel<1>;
    case 2: return mfb_fused_kernel<2>;
    case 3: return mfb_fused_kernel<3>;
    default: return mfb_fused_kernel<4>;
    }
}

static mfb_fwd_fn mfb_fwd_for(const gh_ctx *c)
{
    switch (mfb_kind(c)) {
    case 0: return mfb_forward_kernel<0>;
    case 1: return mfb_forward_kernel<1>;
    case 2: return mfb_forward_kernel<2>;
    case 3: return mfb_forward_kernel<3>;
    default: return mfb_forward_kernel<4>;
    }
}

// Tesseroids with the near-field list: the listed pairs as differences to the root leaf the team / batch
// passes stage for every pair (column-major for the adjoint, a row-major copy for the forward: both sums
// then run in a fixed order without atomics).  Built once per context.
static int mf_near_deltas(gh_ctx *c)
{
    gh_ctx::Batch &b = c->bt;
    if (b.near_built) return GH_OK;
    b.near_built = true;
    b.mfb_near = false;
    const int kind = mfb_kind(c);
    if (kind >= 2 && c->mf_near_n > 0) {
        const int64_t n = c->mf_near_n;
        std::vector<int64_t> ptr((size_t)c->M + 1);
        std::vector<int> row((size_t)n), colof((size_t)n);
        HIPCHK(c, hipMemcpyAsync(ptr.data(), c->mf_near_ptr, sizeof(int64_t) * ptr.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(row.data(), c->mf_near_row, sizeof(int) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int64_t j = 0; j < c->M; ++j)
            for (int64_t q = ptr[(size_t)j]; q < ptr[(size_t)j + 1]; ++q) colof[(size_t)q] = (int)j;
        int *d_colof = nullptr;
        TRY(dalloc(c, &d_colof, (size_t)n, false));
        HIPCHK(c, hipMemcpyAsync(d_colof, colof.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        TRY(dalloc(c, &b.ndelta, (size_t)n, false));
        const MfGeom g = mf_geom(c);
        MfNear near{c->mf_near_ptr, c->mf_near_row, c->mf_near_val};
        if (kind >= 3)
            mfb_near_delta_kernel<3><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(g, c->mf_cellc, near, n, d_colof, b.ndelta);
        else
            mfb_near_delta_kernel<2><<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>(g, c->mf_cellc, near, n, d_colof, b.ndelta);
        HIPCHK(c, hipGetLastError());
        std::vector<double> delta((size_t)n);
        TRY(d2h(c, delta.data(), b.ndelta, (size_t)n));
        // row-major copy (columns ascending inside a row: a stable counting sort by row)
        std::vector<int64_t> rptr((size_t)c->N + 1, 0);
        for (int64_t q = 0; q < n; ++q) rptr[(size_t)row[(size_t)q] + 1] += 1;
        for (int64_t i = 0; i < c->N; ++i) rptr[(size_t)i + 1] += rptr[(size_t)i];
        std::vector<int64_t> fill(rptr.begin(), rptr.end() - 1);
        std::vector<int> rcol((size_t)n);
        std::vector<double> rdelta((size_t)n);
        for (int64_t q = 0; q < n; ++q) {
            const int64_t at = fill[(size_t)row[(size_t)q]]++;
            rcol[(size_t)at] = colof[(size_t)q];
            rdelta[(size_t)at] = delta[(size_t)q];
        }
        TRY(dalloc(c, &b.rptr, (size_t)c->N + 1, false));
        TRY(dalloc(c, &b.rcol, (size_t)n, false));
        TRY(dalloc(c, &b.rdelta, (size_t)n, false));
        HIPCHK(c, hipMemcpyAsync(b.rptr, rptr.data(), sizeof(int64_t) * rptr.size(), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b.rcol, rcol.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b.rdelta, rdelta.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));  // (the host vectors go out of scope)
        TRY(dalloc(c, &b.Snear, (size_t)c->M * CB));
        b.mfb_near = true;
    }
    return GH_OK;
}

// ---- ONE chain on teams of workgroups (mf_team_kernel) -----------------------------------------------
typedef void (*mft_fn)(MfGeom, SweepArgs, MftArgs, const double *, const double *, MfStats *);

static mft_fn mft_for(const gh_ctx *c)
{
    switch (mfb_kind(c)) {
    case 0: return mf_team_kernel<0>;
    case 1: return mf_team_kernel<1>;
    case 2: return mf_team_kernel<2>;
    case 3: return mf_team_kernel<3>;
    default: return mf_team_kernel<4>;
    }
}

static bool mft_plan(gh_ctx *c)
{
    gh_ctx::MfTeam &t = c->mft;
    if (t.state != 0) return t.state > 0;
    t.state = -1;
    if (!c->mf || !c->mf_fused || lonsym_on(c) || c->sh.kind != 0) return false;
    if (c->cell_kind != GH_CELL_TESSEROID || mfb_kind(c) < 2) return false;  // (the fast / reference-order leaf with the near-field list)
    if (env_int("GRAVHMC_MF_TEAM", 1) == 0) return false;
    const int64_t ntiles = (c->M + 15) / 16;
    const int nrb = (int)((c->ld + 63) / 64);
    t.members = (nrb + MFB_RC_FUS - 1) / MFB_RC_FUS;
    if (t.members > MFT_MAXMEM || t.members > c->cus) return false;
    int fr = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, c->cus / t.members));
    fr = std::min(fr, c->grid - 1);  // (slab rows: one per range and one for the near field)
    if (fr < 1) return false;
    t.tpr = (int)((ntiles + fr - 1) / fr);
    t.ranges = (int)((ntiles + t.tpr - 1) / t.tpr);
    mft_fn f = mft_for(c);
    int per_cu = 0;
    if (allow_dynamic_lds(reinterpret_cast<const void *>(f), MFT_LDS) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(f), 1024, MFT_LDS) != hipSuccess ||
        per_cu < 1 || (int64_t)per_cu * c->cus < (int64_t)t.members * t.ranges) {
        (void)hipGetLastError();
        return false;
    }
    if (mf_near_deltas(c) != GH_OK) return false;
    if (dalloc(c, &t.gran, (size_t)t.ranges * MFB_FUS_RING * MFT_MAXMEM * 32) != GH_OK || dalloc(c, &t.abort_w, 4) != GH_OK ||
        dalloc(c, &t.snear, (size_t)c->M) != GH_OK)
        return false;
    t.tag = 0;
    t.state = 1;
    return true;
}

// the fused leapfrog pass of one chain on teams (modes with SW_ADJ and an update / final half step)
static int mft_launch(gh_ctx *c, SweepArgs &a)
{
    gh_ctx::MfTeam &t = c->mft;
    gh_ctx::Batch &b = c->bt;
    a.ld = c->ld;
    a.M = c->M;
    if ((uint64_t)t.tag + (uint64_t)t.tpr + 2 > 0xf0000000ull) {
        HIPCHK(c, hipMemsetAsync(t.gran, 0, sizeof(u64) * (size_t)t.ranges * MFB_FUS_RING * MFT_MAXMEM * 32, c->stream));
        t.tag = 0;
    }
    const double *wm = c->weighted ? c->wm : nullptr;
    if (b.mfb_near)
        mf1_near_adjoint_kernel<<<dim3((unsigned)c->M), dim3(64), 0, c->stream>>>(c->mf_near_ptr, c->mf_near_row, b.ndelta,
                                                                                 a.r, t.snear);
    MftArgs f;
    f.tiles_per_range = t.tpr;
    f.gran = t.gran;
    f.tag0 = t.tag;
    f.abort_w = t.abort_w;
    f.poll_members = t.members + ((env_int("GRAVHMC_MF_TEAM_TEST_ABORT", 0) && t.members < MFT_MAXMEM) ? 1 : 0);
    f.n_pp = c->n_teams;
    f.snear = b.mfb_near ? t.snear : nullptr;
    hipLaunchKernelGGL(mft_for(c), dim3((unsigned)t.members, (unsigned)t.ranges), dim3(1024), MFT_LDS, c->stream, mf_geom(c),
                       a, f, wm, c->mf_cellc, c->prof ? c->mf_stats : nullptr);
    t.tag += (unsigned)t.tpr + 1u;
    t.inflight = true;
    t.launches += 1;
    if (a.mode & SW_FWD) {
        int rows = t.ranges;
        if (b.mfb_near) {
            const double *x = (a.mode & SW_UPD) ? a.x_out : a.x_in;
            HIPCHK(c, hipMemsetAsync(a.slab + (size_t)rows * (size_t)c->ld, 0, sizeof(double) * (size_t)c->ld, c->stream));
            mf1_near_forward_kernel<<<dim3((unsigned)c->N), dim3(64), 0, c->stream>>>(b.rptr, b.rcol, b.rdelta, c->N, x, wm,
                                                                                     a.slab + (size_t)rows * (size_t)c->ld);
            rows += 1;
        }
        c->slab_live = rows;
    }
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

// after a synchronisation point: did a team pass of the single chain give up?  (Then everything it fed
// is void; the teams are off for good, the caller repeats its trajectory on the column-per-workgroup pass.)
static int mft_failed(gh_ctx *c, bool *failed)
{
    gh_ctx::MfTeam &t = c->mft;
    *failed = false;
    if (!t.inflight) return GH_OK;
    t.inflight = false;
    unsigned w[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(w, t.abort_w, sizeof w, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (w[0] == 0u) return GH_OK;
    fprintf(stderr, "libgravhmc: the matrix-free team pass timed out waiting for its workgroups; repeating the "
                    "trajectory on the column-per-workgroup pass\n");
    HIPCHK(c, hipMemsetAsync(t.abort_w, 0, 4 * sizeof(unsigned), c->stream));
    t.state = -1;
    t.aborts += 1;
    *failed = true;
    return GH_OK;
}

// Partition of the two passes and, for tesseroids with the near-field table, the listed pairs as
// differences to the root leaf the dense passes stage (column-major for the adjoint, a row-major
// copy for the forward: both sums then run in a fixed order without atomics).
static int mfb_plan(gh_ctx *c)
{
    gh_ctx::Batch &b = c->bt;
    const int64_t ntiles = (c->M + 15) / 16;
    HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(mfb_adj_for(c)), MFB_LDS_ADJ));
    HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(mfb_fwd_for(c)), MFB_LDS_FWD));
    // one workgroup of 16 waves per CU (133 KB of staging): adjoint = column tiles dealt round-robin
    b.mfb_grid_adj = (int)std::min<int64_t>(ntiles, (int64_t)c->cus * env_int("GRAVHMC_MFB_WG_PER_CU", 1));
    b.n_waves = b.mfb_grid_adj;  // rows of pp_part: one per workgroup
    // forward = 512-row chunks x ranges of column tiles, about one workgroup per CU
    const int nrb = (int)((c->ld + 63) / 64);
    b.mfb_rchunks = (nrb + MFB_RC_FWD - 1) / MFB_RC_FWD;
    int ranges = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, c->cus / b.mfb_rchunks));
    ranges = env_int("GRAVHMC_MFB_RANGES", ranges);
    b.mfb_tpr = (int)((ntiles + ranges - 1) / ranges);
    b.mfb_ranges = (int)((ntiles + b.mfb_tpr - 1) / b.mfb_tpr);
    TRY(dalloc(c, &b.iw, (size_t)c->M));
    // one evaluation per entry and step: teams of workgroups (mfb_fused_kernel), every workgroup resident
    b.fus_on = false;
    b.fus_members = (nrb + MFB_RC_FUS - 1) / MFB_RC_FUS;
    if (env_int("GRAVHMC_MFB_FUSED", 1) != 0 && b.fus_members <= MFB_FUS_MAXMEM && b.fus_members <= c->cus) {
        int fr = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, c->cus / b.fus_members));
        b.fus_tpr = (int)((ntiles + fr - 1) / fr);
        b.fus_ranges = (int)((ntiles + b.fus_tpr - 1) / b.fus_tpr);
        mfb_fus_fn ff = mfb_fus_for(c);
        int per_cu = 0;
        if (allow_dynamic_lds(reinterpret_cast<const void *>(ff), MFB_LDS_FUS) == hipSuccess &&
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(ff), 1024, MFB_LDS_FUS) ==
                hipSuccess &&
            per_cu >= 1 && (int64_t)per_cu * c->cus >= (int64_t)b.fus_members * b.fus_ranges) {
            TRY(dalloc(c, &b.fus_gran, (size_t)b.fus_ranges * MFB_FUS_RING * MFB_FUS_MAXMEM * 512));
            TRY(dalloc(c, &b.fus_abort, 4));
            b.fus_tag = 0;
            b.fus_on = true;
        } else {
            (void)hipGetLastError();
        }
    }
    TRY(mf_near_deltas(c));
    b.n_colblocks = std::max(b.mfb_ranges, b.fus_on ? b.fus_ranges : 0) + (b.mfb_near ? 1 : 0);
    b.slab_live = b.n_colblocks;
    b.cols_per_block = (int64_t)b.mfb_tpr * 16;
    return GH_OK;
}

// The adjoint GEMM of the two-pass batch wants G in MFMA operand order; 288 GB of HBM usually has room for the
// second copy (C2: 40 GB + 40 GB).  Without it the kernel reads the column-major matrix.
static int batch_relayout(gh_ctx *c)
{
    gh_ctx::Batch &b = c->bt;
    if (c->mf || b.Gb || !c->G || !env_int("GRAVHMC_BATCH_RELAYOUT", 1)) return GH_OK;
    const int64_t ntiles = (c->M + 15) / 16;
    size_t free_b = 0, total_b = 0;
    const size_t need_b = sizeof(double) * (size_t)ntiles * 16 * (size_t)c->ld;
    if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need_b + ((size_t)2 << 30)) {
        void *ptr = nullptr;
        if (hipMalloc(&ptr, need_b) == hipSuccess) {
            c->allocs.push_back(ptr);
            b.Gb = static_cast<double *>(ptr);
            batch_relayout_kernel<<<dim3(1 << 16), dim3(256), 0, c->stream>>>(c->G, c->ld, c->M, (int)(c->ld / 16), ntiles, b.Gb);
            HIPCHK(c, hipGetLastError());
        } else {
            (void)hipGetLastError();
        }
    }
    return GH_OK;
}

// Stored kernel: teams of workgroups that read G once per step (batch_team_kernel).  Needs every workgroup
// resident (one per CU) and 8 .. 32 members (3137 .. 14336 rows); otherwise the two-pass kernels stay.
static int bteam_plan(gh_ctx *c)
{
    gh_ctx::Batch &b = c->bt;
    b.fus_on = false;
    if (!c->G) return GH_OK;
    const int64_t ntiles = (c->M + 15) / 16;
    const int nrb = (int)((c->ld + 63) / 64);
    b.fus_members = (nrb + BT_RC - 1) / BT_RC;
    if (c->ld % 16 != 0 || b.fus_members < BT_MINMEM || b.fus_members > BT_MAXMEM || b.fus_members > c->cus) return GH_OK;
    const int fr = (int)std::max<int64_t>(1, std::min<int64_t>(ntiles, c->cus / b.fus_members));
    b.fus_tpr = (int)((ntiles + fr - 1) / fr);
    b.fus_ranges = (int)((ntiles + b.fus_tpr - 1) / b.fus_tpr);
    b.fus_nval = (256 + b.fus_members - 1) / b.fus_members;
    const void *fn = reinterpret_cast<const void *>(batch_team_kernel);
    int per_cu = 0;
    if (allow_dynamic_lds(fn, BT_LDS) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, BT_NW * 64, BT_LDS) != hipSuccess || per_cu < 1 ||
        (int64_t)per_cu * c->cus < (int64_t)b.fus_members * b.fus_ranges) {
        (void)hipGetLastError();
        return GH_OK;
    }
    TRY(dalloc(c, &b.fus_gran, (size_t)b.fus_ranges * BT_RING * BT_MAXMEM * 512));
    TRY(dalloc(c, &b.fus_granx, (size_t)b.fus_ranges * BT_RING * 512));
    TRY(dalloc(c, &b.fus_abort, 4));
    b.fus_tag = 0;
    b.fus_on = true;
    return GH_OK;
}

static int batch_time_begin(gh_ctx *c, bool &timed);
static int batch_time_end(gh_ctx *c, bool timed);

// forward of all chains at X into the slab (ranges of column tiles, plus the near-field block)
static int mfb_forward(gh_ctx *c, const double *X)
{
    gh_ctx::Batch &b = c->bt;
    int ranges = b.mfb_ranges;
    if (b.fus_fwd_of == X) {
        // the fused kernel that produced X left its forward partials in the slab already
        ranges = b.fus_ranges;
    } else {
        MfbFwdArgs f;
        f.ld = c->ld;
        f.M = c->M;
        f.X = X;
        f.iw = b.iw;
        f.tiles_per_range = b.mfb_tpr;
        f.slab = b.slab;
        f.dbg = env_int("GRAVHMC_MFB_DBG", 0);
        bool timed;
        TRY(batch_time_begin(c, timed));
        hipLaunchKernelGGL(mfb_fwd_for(c), dim3((unsigned)b.mfb_rchunks, (unsigned)b.mfb_ranges), dim3(1024), MFB_LDS_FWD,
                           c->stream, mf_geom(c), f, c->cell_kind == GH_CELL_TESSEROID ? c->mf_cellc : nullptr,
                           c->prof ? c->mf_stats : nullptr);
        TRY(batch_time_end(c, timed));
        if (c->prof) c->mf_launches += 1;
    }
    b.fus_fwd_of = nullptr;
    if (b.mfb_near) {
        const int64_t l16 = c->ld * CB;
        mfb_near_forward_kernel<<<dim3((unsigned)c->ld), dim3(256), 0, c->stream>>>(
            b.rptr, b.rcol, b.rdelta, c->N, c->ld, X, b.iw, b.slab + (size_t)ranges * (size_t)l16);
    }
    b.slab_live = ranges + (b.mfb_near ? 1 : 0);
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

// adjoint of all chains + leapfrog update: the dense batch's MFMA kernel or the matrix-free one.
// fwd_follows: the caller evaluates the potential at a.X_out next (batch_evaluate) -- on the matrix-free
// kernel the fused team pass then delivers the forward partials from the same evaluation of the entries.
static int batch_launch_adjoint(gh_ctx *c, BatchAdjArgs &a, bool fwd_follows)
{
    gh_ctx::Batch &b = c->bt;
    bool timed;
    b.fus_fwd_of = nullptr;
    if (c->mf && b.fus_on && fwd_follows) {
        if (b.mfb_near) {
            mfb_near_adjoint_kernel<<<dim3((unsigned)c->M), dim3(256), 0, c->stream>>>(
                c->mf_near_ptr, c->mf_near_row, b.ndelta, c->M, a.Rt, b.Snear);
        }
        if ((uint64_t)b.fus_tag + (uint64_t)b.fus_tpr + 2 > 0xf0000000ull) {
            HIPCHK(c, hipMemsetAsync(b.fus_gran, 0, sizeof(u64) * (size_t)b.fus_ranges * MFB_FUS_RING * MFB_FUS_MAXMEM * 512,
                                     c->stream));
            b.fus_tag = 0;
        }
        MfbFusArgs f;
        f.tiles_per_range = b.fus_tpr;
        f.slab = b.slab;
        f.gran = b.fus_gran;
        f.tag0 = b.fus_tag;
        f.abort_w = b.fus_abort;
        // test hook: the members wait for a part that never comes, time out and give up
        f.poll_members = b.fus_members + ((env_int("GRAVHMC_MFB_TEST_ABORT", 0) && b.fus_members < MFB_FUS_MAXMEM) ? 1 : 0);
        f.n_pp = b.n_waves;
        f.dbg = nullptr;
        if (env_int("GRAVHMC_MFB_TIMING", 0)) {
            TRY(dalloc(c, &b.fus_dbg, 8));
            f.dbg = b.fus_dbg;
        }
        TRY(batch_time_begin(c, timed));
        hipLaunchKernelGGL(mfb_fus_for(c), dim3((unsigned)b.fus_members, (unsigned)b.fus_ranges), dim3(1024), MFB_LDS_FUS,
                           c->stream, mf_geom(c), a, f, b.iw, c->cell_kind == GH_CELL_TESSEROID ? c->mf_cellc : nullptr,
                           b.mfb_near ? b.Snear : nullptr, c->prof ? c->mf_stats : nullptr);
        TRY(batch_time_end(c, timed));
        if (c->prof) c->mf_launches += 1;
        b.fus_tag += (unsigned)b.fus_tpr + 1u;
        b.fus_inflight = true;
        b.fus_launches += 1;
        b.fus_fwd_of = a.X_out;
    } else if (c->mf) {
        if (b.mfb_near) {
            mfb_near_adjoint_kernel<<<dim3((unsigned)c->M), dim3(256), 0, c->stream>>>(
                c->mf_near_ptr, c->mf_near_row, b.ndelta, c->M, a.Rt, b.Snear);
        }
        TRY(batch_time_begin(c, timed));
        hipLaunchKernelGGL(mfb_adj_for(c), dim3((unsigned)b.mfb_grid_adj), dim3(1024), MFB_LDS_ADJ, c->stream, mf_geom(c),
                           a, b.iw, c->cell_kind == GH_CELL_TESSEROID ? c->mf_cellc : nullptr,
                           b.mfb_near ? b.Snear : nullptr, c->prof ? c->mf_stats : nullptr);
        TRY(batch_time_end(c, timed));
        if (c->prof) c->mf_launches += 1;
    } else if (b.fus_on) {
        // stored kernel on teams: adjoint of all chains, update and the forward at the new positions from ONE
        // read of G (an adjoint nothing follows costs the same read: the forward rides along unused)
        if ((uint64_t)b.fus_tag + (uint64_t)b.fus_tpr + 2 > 0xf0000000ull) {
            HIPCHK(c, hipMemsetAsync(b.fus_gran, 0, sizeof(u64) * (size_t)b.fus_ranges * BT_RING * BT_MAXMEM * 512, c->stream));
            HIPCHK(c, hipMemsetAsync(b.fus_granx, 0, sizeof(u64) * (size_t)b.fus_ranges * BT_RING * 512, c->stream));
            b.fus_tag = 0;
        }
        BtArgs f;
        f.tiles_per_range = b.fus_tpr;
        f.nval = b.fus_nval;
        f.slab = b.slab;
        f.gran_p = b.fus_gran;
        f.gran_x = b.fus_granx;
        f.tag0 = b.fus_tag;
        f.abort_w = b.fus_abort;
        // test hook: the members wait for a part that never comes, time out and give up
        f.poll_members = b.fus_members + ((env_int("GRAVHMC_BATCH_TEAM_TEST_ABORT", 0) && b.fus_members < BT_MAXMEM) ? 1 : 0);
        f.n_pp = b.n_waves;
        f.dbg = nullptr;
        if (env_int("GRAVHMC_MFB_TIMING", 0)) {
            TRY(dalloc(c, &b.fus_dbg, 8));
            f.dbg = b.fus_dbg;
        }
        f.dbg_mem = env_int("GRAVHMC_BT_DBG_MEM", 0);
        f.dbg_wave = env_int("GRAVHMC_BT_DBG_WAVE", 0);
        f.dbg_break = env_int("GRAVHMC_BT_BREAK", 0);
        TRY(batch_time_begin(c, timed));
        hipLaunchKernelGGL(batch_team_kernel, dim3((unsigned)b.fus_members, (unsigned)b.fus_ranges), dim3(BT_NW * 64), BT_LDS,
                           c->stream, a, f);
        TRY(batch_time_end(c, timed));
        b.fus_tag += (unsigned)b.fus_tpr + 1u;
        b.fus_inflight = true;
        b.fus_launches += 1;
        if (fwd_follows) b.fus_fwd_of = a.X_out;
    } else {
        TRY(batch_time_begin(c, timed));
        batch_adjoint_kernel<<<dim3((unsigned)(b.n_waves / 4)), dim3(256), 0, c->stream>>>(a);
        TRY(batch_time_end(c, timed));
    }
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

static int batch_alloc(gh_ctx *c)
{
    gh_ctx::Batch &b = c->bt;
    const size_t M16 = (size_t)c->M * CB, L16 = (size_t)c->ld * CB;
    if (b.Xc) return GH_OK;
    TRY(dalloc(c, &b.Xc, M16));
    TRY(dalloc(c, &b.Rtc, L16));
    TRY(dalloc(c, &b.GREGc, M16));
    TRY(dalloc(c, &b.Dc, L16));
    for (int i = 0; i < 2; ++i) {
        TRY(dalloc(c, &b.Xw[i], M16));
        TRY(dalloc(c, &b.Pw[i], M16));
    }
    TRY(dalloc(c, &b.Rtw, L16));
    TRY(dalloc(c, &b.GREGw, M16));
    TRY(dalloc(c, &b.Dw, L16));
    TRY(dalloc(c, &b.scal, CB * 4));
    TRY(dalloc(c, &b.stage, M16));
    const int64_t ntiles = (c->M + 15) / 16;
    if (c->mf) {
        TRY(mfb_plan(c));
    } else {
        // forward: 512-row blocks x column blocks, about 4 workgroups per CU in total
        const int rowblocks = (int)((c->ld + 511) / 512);
        int colblocks = std::max(1, (c->cus * 4 + rowblocks - 1) / rowblocks);
        int64_t cpb = (c->M + colblocks - 1) / colblocks;
        cpb = (cpb + 15) / 16 * 16;
        b.cols_per_block = cpb;
        b.n_colblocks = (int)((c->M + cpb - 1) / cpb);
        const int64_t npairs = (ntiles + 1) / 2;  // a wave owns two adjacent column tiles
        const int wgs = (int)std::min<int64_t>((npairs + 3) / 4, (int64_t)c->cus * 4);
        b.n_waves = wgs * 4;
        // One read of G per step on teams (batch_team_kernel) where the problem fits them, else -- or with
        // GRAVHMC_BATCH_TEAM=0 -- two reads and a second, operand-ordered copy of G.  Measured at C2 with 16
        // chains: 1 286.6 chain-steps/s on the teams against 1 187.0 in the committed driver-command record of round 3
        // (profiles/r03/bench_c2_driver_command.json: +8.4 %, sampler level; the launches alone 11.1 ... 11.6 ms against
        // 6.5 + 6.5), with 40 GB of HBM instead of 80.
        if (env_int("GRAVHMC_BATCH_TEAM", 1) != 0) TRY(bteam_plan(c));
        if (b.fus_on) {
            b.n_waves = std::max(b.n_waves, (b.fus_members * b.fus_ranges + 3) / 4 * 4);  // rows of pp_part
        }
    }
    TRY(dalloc(c, &b.slab, (size_t)std::max(b.n_colblocks, b.fus_on ? b.fus_ranges : 0) * L16));
    b.n_regblocks = (int)((c->M + 15) / 16);
    TRY(dalloc(c, &b.regpart, (size_t)b.n_regblocks * CB));
    TRY(dalloc(c, &b.pp_part, (size_t)b.n_waves * CB));
    b.n_pp0 = (int)std::min<int64_t>(512, (c->M + 15) / 16);
    TRY(dalloc(c, &b.pp0_part, (size_t)b.n_pp0 * CB));
    HIPCHK(c, hipHostMalloc((void **)&b.h, sizeof(double) * (size_t)(CB * 4 + (b.n_waves + 2 * b.n_pp0) * CB)));
    // (the two-pass batch's second copy of G: not with the team pass -- if that ever gives up, the copy is made then)
    if (!b.fus_on) TRY(batch_relayout(c));
    TRY(dalloc(c, &c->tmpM, (size_t)c->M));
    TRY(dalloc(c, &c->low, (size_t)c->M));
    TRY(dalloc(c, &c->high, (size_t)c->M));
    if (!c->mwapr) TRY(dalloc(c, &c->mwapr, (size_t)c->M));
    if (!c->wm2) TRY(dalloc(c, &c->wm2, (size_t)c->M));
    return GH_OK;
}

static int batch_time_begin(gh_ctx *c, bool &timed)
{
    timed = c->prof && c->ev_used + 2 <= c->ev.size();
    if (timed) HIPCHK(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    return GH_OK;
}

static int batch_time_end(gh_ctx *c, bool timed)
{
    if (timed) {
        HIPCHK(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        c->ev_bytes[c->ev_used / 2] = c->N * c->M * (int64_t)sizeof(double);
        c->ev_used += 2;
    }
    c->bt.sweeps += 1;
    return GH_OK;
}

// After a synchronisation point: did a fused team pass since the last look give up (its workgroups were
// not all resident -- the GPU shared with somebody else)?  Everything it fed is void; the fused form
// is switched off for good (the two-pass kernels need no co-residency).
static int mfb_fused_failed(gh_ctx *c, bool *failed)
{
    gh_ctx::Batch &b = c->bt;
    *failed = false;
    if (!b.fus_inflight) return GH_OK;
    b.fus_inflight = false;
    unsigned w[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(w, b.fus_abort, sizeof w, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (w[0] == 0u) return GH_OK;
    fprintf(stderr, "libgravhmc: the %s timed out waiting for its workgroups; continuing with the two-pass kernels\n",
            c->mf ? "fused matrix-free batch pass" : "batch's team pass");
    HIPCHK(c, hipMemsetAsync(b.fus_abort, 0, 4 * sizeof(unsigned), c->stream));
    b.fus_on = false;
    b.fus_aborts += 1;
    b.fus_fwd_of = nullptr;
    *failed = true;
    TRY(batch_relayout(c));  // (stored kernel: the two-pass adjoint's copy of G, not made while the teams ran)
    return GH_OK;
}

// forward of all chains at X, then regulariser and residuals into (D, GREG, Rt, scal)
static int batch_evaluate(gh_ctx *c, const double *X, double *D, double *GREG, double *Rt, double *scal = nullptr)
{
    gh_ctx::Batch &b = c->bt;
    bool timed;
    int nblocks = b.n_colblocks;
    if (c->mf) {
        TRY(mfb_forward(c, X));
    } else if (b.fus_fwd_of == X) {
        // the team pass that produced X left its forward partials in the slab already
        b.fus_fwd_of = nullptr;
        nblocks = b.fus_ranges;
    } else {
        b.fus_fwd_of = nullptr;
        BatchFwdArgs f;
        f.G = c->G;
        f.ld = c->ld;
        f.M = c->M;
        f.N = c->N;
        f.X = X;
        f.cols_per_block = b.cols_per_block;
        f.slab = b.slab;
        TRY(batch_time_begin(c, timed));
        batch_forward_kernel<<<dim3((unsigned)((c->ld + 511) / 512), (unsigned)b.n_colblocks), dim3(256), 0,
                               c->stream>>>(f);
        TRY(batch_time_end(c, timed));
    }
    const int64_t n16 = c->ld * CB;
    batch_reduce_kernel<<<dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream>>>(
        b.slab, c->mf ? b.slab_live : nblocks, n16, D);
    BatchRegArgs ra;
    ra.kind = c->reg_kind;
    ra.M = c->M;
    ra.nz = c->shape[0];
    ra.ny = c->shape[1];
    ra.nx = c->shape[2];
    ra.alpha = c->alpha;
    ra.beta = c->beta;
    ra.X = X;
    ra.mwapr = c->mwapr;
    ra.wm2 = c->wm2;
    ra.GREG = GREG;
    ra.regpart = b.regpart;
    batch_reg_kernel<<<dim3((unsigned)b.n_regblocks), dim3(256), 0, c->stream>>>(ra);
    BatchFinishArgs fa;
    fa.N = c->N;
    fa.ld = c->ld;
    fa.n_regpart = b.n_regblocks;
    fa.D = D;
    fa.gfix = c->have_fix ? c->gfix : nullptr;
    fa.dobs_c = c->dobs_c;
    fa.regpart = b.regpart;
    fa.alpha = c->alpha;
    fa.Rt = Rt;
    fa.scal = scal ? scal : b.scal;
    batch_finish_kernel<<<dim3(CB), dim3(1024), 0, c->stream>>>(fa);
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

static int batch_upload_rows(gh_ctx *c, const double *rows, int C, double *dst)
{
    gh_ctx::Batch &b = c->bt;
    HIPCHK(c, hipMemcpyAsync(b.stage, rows, sizeof(double) * (size_t)C * (size_t)c->M, hipMemcpyHostToDevice,
                             c->stream));
    const int64_t n16 = c->M * CB;
    batch_interleave_kernel<<<dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream>>>(b.stage, C, c->M, dst);
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

// state of the MFMA batch (chain-interleaved layouts) at the models x0s (C rows of M)
static int batch_init_mfma(gh_ctx *c, int C, const double *x0s)
{
    TRY(batch_alloc(c));
    gh_ctx::Batch &b = c->bt;
    b.C = C;
    if (c->mf) {
        // 1 / wm as the single-chain passes round it (x * (1.0 / w); 1 where w == 0 or not weighted)
        mfb_invw_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(
            c->weighted ? c->wm : nullptr, c->M, b.iw);
        HIPCHK(c, hipGetLastError());
    }
    TRY(batch_upload_rows(c, x0s, C, b.Xc));
    TRY(batch_evaluate(c, b.Xc, b.Dc, b.GREGc, b.Rtc));
    TRY(d2h(c, b.h, b.scal, CB * 4));
    for (int k = 0; k < CB; ++k) {
        b.U[k][0] = b.h[4 * k + 2];
        b.U[k][1] = b.h[4 * k + 0];
        b.U[k][2] = b.h[4 * k + 1];
    }
    b.ready = true;
    return GH_OK;
}
