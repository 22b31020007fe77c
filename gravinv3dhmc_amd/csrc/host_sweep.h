// libgravhmc host side: choice of the sweep instantiation, column partition, launches (dense,
// row panels, matrix-free).  Included once by gravhmc.hip.
#pragma once

// ----------------------------------------------------------------- sweep dispatch

typedef void (*sweep_fn)(SweepArgs);
typedef void (*weight_fn)(double *, int64_t, int64_t, int64_t, int, double, double *);

template <int TW, int PF, bool NT>
static sweep_fn pick_sweep_e(int ept2)
{
    switch (ept2) {
    case 1: return sweep_kernel<TW, 1, PF, NT>;
    case 2: return sweep_kernel<TW, 2, PF, NT>;
    case 3: return sweep_kernel<TW, 3, PF, NT>;
    case 4: return sweep_kernel<TW, 4, PF, NT>;
    case 5: return sweep_kernel<TW, 5, PF, NT>;
    case 6: return sweep_kernel<TW, 6, PF, NT>;
    case 8: return sweep_kernel<TW, 8, PF, NT>;
    }
    return nullptr;
}

template <int TW>
static sweep_fn pick_sweep(int ept2, int pf, bool nt)
{
    if (pf == 2) return nt ? pick_sweep_e<TW, 2, true>(ept2) : pick_sweep_e<TW, 2, false>(ept2);
    return nt ? pick_sweep_e<TW, 1, true>(ept2) : pick_sweep_e<TW, 1, false>(ept2);
}

template <int TW>
static weight_fn pick_weight(int ept2)
{
    switch (ept2) {
    case 1: return weight_kernel<TW, 1>;
    case 2: return weight_kernel<TW, 2>;
    case 3: return weight_kernel<TW, 3>;
    case 4: return weight_kernel<TW, 4>;
    case 5: return weight_kernel<TW, 5>;
    case 6: return weight_kernel<TW, 6>;
    case 8: return weight_kernel<TW, 8>;
    }
    return nullptr;
}

static sweep_fn sweep_for(const gh_ctx *c)
{
    if (c->TW == 1) return pick_sweep<1>(c->EPT2, c->PF, c->NT);
    if (c->TW == 4) return pick_sweep<4>(c->EPT2, c->PF, c->NT);
    if (c->TW == 8) return pick_sweep<8>(c->EPT2, c->PF, c->NT);
    return pick_sweep<16>(c->EPT2, c->PF, c->NT);
}

static weight_fn weight_for(const gh_ctx *c)
{
    if (c->TW == 1) return pick_weight<1>(c->EPT2);
    if (c->TW == 4) return pick_weight<4>(c->EPT2);
    if (c->TW == 8) return pick_weight<8>(c->EPT2);
    return pick_weight<16>(c->EPT2);
}

static int env_int(const char *name, int dflt)
{
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// hipFuncAttributeMaxDynamicSharedMemorySize belongs to the kernel, not to a context: several
// contexts of one process share an instantiation, so the allowance is only ever raised.
static hipError_t allow_dynamic_lds(const void *func, size_t bytes)
{
    static std::mutex mu;
    static std::map<const void *, size_t> allowed;
    std::lock_guard<std::mutex> lock(mu);
    size_t &cur = allowed[func];
    if (bytes <= cur) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) cur = bytes;
    return e;
}

// Choose team width / registers per thread from ld, and the column partition from M.
static int configure_sweep(gh_ctx *c)
{
    const int64_t ld = c->ld;
    int tw, per;  // rows one unit of EPT2 covers = tw*64*2
    c->n_panels = 1;
    c->panel_rows = ld;
    if (ld <= 1024) tw = 1;
    else if (ld <= 4096) tw = 4;
    // 8-wave teams, two per CU, while a thread holds <= 6 double2 (measured at N = 5000 / 6000:
    // 6.5 / 6.2 TB/s against 5.2 / 5.8 with 16-wave teams; at 7381 and 8192 rows 16 waves win)
    else if (ld <= 6144 && env_int("GRAVHMC_TW8", 1)) tw = 8;
    else if (ld <= 16384) tw = 16;
    else {
        // more rows than a team can hold in registers: row panels of <= 16384 rows.  The dot
        // product of a column then spans several launches, so the adjoint and the forward can no
        // longer share one read of G (two reads per step, like the reference's formulation).
        tw = 16;
        c->n_panels = (int)((ld + 10239) / 10240);  // <= 10240 rows: 5 double2 per thread, no spills
        c->panel_rows = ((ld + c->n_panels - 1) / c->n_panels + 15) / 16 * 16;
    }
    {
        // diagnostic override of the team width (must still hold a column: 8 double2 per thread)
        const int tw_env = env_int("GRAVHMC_TW", 0);
        if ((tw_env == 1 || tw_env == 4 || tw_env == 8 || tw_env == 16) && (int64_t)tw_env * 1024 >= c->panel_rows)
            tw = tw_env;
    }
    per = tw * 128;
    int e = (int)((c->panel_rows + per - 1) / per);
    if (e == 7) e = 8;
    c->TW = tw;
    c->EPT2 = e;
    // columns in flight per team beyond the one being reduced.  Re-measured once the requests really
    // stayed in flight (kernels.hip.h: sweep_kernel, load_col): one is enough wherever it was tried
    // (7381 rows, 4 double2: 6.6 TB/s against 6.2 with two; 8192 / 12000 / 16000 rows: 6.46 / 6.47 /
    // 6.45; two at 12000 / 16000: 6.1 / 3.8, spills); at C2's 5 double2 two measure 0.5 % better
    // (122 VGPRs) and are kept.
    c->PF = env_int("GRAVHMC_PF", (tw == 16 && e == 5) ? 2 : 1) == 2 ? 2 : 1;
    // G larger than the Infinity Cache is streamed once per sweep: bypass-friendly loads
    c->NT = env_int("GRAVHMC_NT", c->ld * c->M * 8 > (int64_t)(512 << 20) ? 1 : 0) != 0;
    const int wg_teams = (tw == 1) ? 4 : 1;
    // resident workgroups per CU we size the grid for (register/LDS budget of the kernel)
    int wg_per_cu = (tw == 16) ? 1 : (tw == 8) ? 2 : 4;
    if (tw == 4) {
        // 4-wave teams (1024 < N <= 4096): as many teams as keep ~8 MB of columns in flight, not
        // more -- every further team costs a slab row per sweep and shortens the teams' column runs
        // (TB/s with 8 / 16 MB at 1500, 2000, 2500, 3000, 4000 rows x 20000 .. 30000 columns:
        // 5.8 / 5.9, 5.0 / 5.1, 5.4 / 5.0, 6.1 / 5.8, 6.4 / 5.8)
        const int64_t col_bytes = c->panel_rows * (int64_t)sizeof(double);
        const int64_t need = (((int64_t)env_int("GRAVHMC_INFLIGHT_MB", 8) << 20) + col_bytes - 1) / col_bytes;
        wg_per_cu = (int)std::max<int64_t>(1, std::min<int64_t>(4, (need + c->cus - 1) / c->cus));
    }
    // never more than the kernel's real residency (registers, LDS): a grid sized for four
    // workgroups per CU of which three fit runs a quarter of its blocks in a second, thin wave
    // (one-wave teams with 8 double2, 140 VGPRs: 5.1 TB/s; sized for the three that fit: 6.5)
    c->lds_bytes = (size_t)(tw == 1 ? 5 * ld : c->panel_rows + 2 * (tw + 8)) * sizeof(double);
    if (c->lds_bytes > 160 * 1024) return fail(c, GH_ERR_UNSUPPORTED, "LDS budget exceeded");
    sweep_fn f = sweep_for(c);
    if (!f) return fail(c, GH_ERR_UNSUPPORTED, "no sweep instantiation for EPT2=%d", e);
    HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(f), c->lds_bytes));
    {
        int occ = 0;
        const int threads = (tw == 1 ? 4 : tw) * 64;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(f), threads,
                                                         c->lds_bytes) == hipSuccess && occ >= 1)
            wg_per_cu = std::min(wg_per_cu, occ);
        else
            (void)hipGetLastError();
    }
    wg_per_cu = env_int("GRAVHMC_WG_PER_CU", wg_per_cu);
    int64_t max_teams = (int64_t)c->cus * wg_per_cu * wg_teams;
    int64_t min_cols = env_int("GRAVHMC_MIN_COLS", tw == 1 ? 2 : 1);
    int64_t cpt = (c->M + max_teams - 1) / max_teams;
    if (cpt < min_cols) cpt = min_cols;
    c->cols_per_team = cpt;
    c->n_teams = (int)((c->M + cpt - 1) / cpt);
    c->n_teams_sweep = c->n_teams;
    c->grid = (c->n_teams + wg_teams - 1) / wg_teams;
    // (one-wave teams interleave inside their block's column range: every wave of the grid is a team)
    if (tw == 1) c->n_teams = c->n_teams_sweep = c->grid * wg_teams;
    // (row panels: partials of vec_update; the team sweep: one per team, at most 128 teams)
    if (c->n_panels > 1) c->n_teams = std::max(std::max(c->n_teams, 128), (int)((c->M + 255) / 256));
    return GH_OK;
}

static MfGeom mf_geom(const gh_ctx *c)
{
    MfGeom g;
    g.kind = c->cell_kind;
    g.radius_u = (c->cell_kind == GH_CELL_TESSEROID && c->obs_h_uniform) ? 6378137.0 + c->obs_h0 : 0.0;
    g.N = c->N;
    g.M = c->M;
    if (c->cell_kind == GH_CELL_TESSEROID) {
        g.o0 = c->tconv;
        g.o1 = c->tconv + c->N;
        g.o2 = c->tconv + 2 * c->N;
        g.o3 = c->tconv + 3 * c->N;
        g.o4 = c->tconv + 4 * c->N;
        g.o5 = c->tconv + 5 * c->N;
    } else {
        g.o0 = c->obs[0];
        g.o1 = c->obs[1];
        g.o2 = c->obs[2];
        g.o3 = nullptr;
        g.o4 = g.o5 = nullptr;
    }
    g.bounds6 = c->bounds;
    g.ratio = c->ratio;
    return g;
}

typedef void (*mf_fused_fn)(MfGeom, SweepArgs, const double *, const double *, MfNear, MfStats *);

template <int KIND>
static mf_fused_fn mf_fused_for_kind(int T, int ept)
{
    if (T == 256) return ept <= 4 ? mf_fused_kernel<256, 4, KIND> : mf_fused_kernel<256, 8, KIND>;
    if (T == 512) return ept <= 8 ? mf_fused_kernel<512, 8, KIND> : mf_fused_kernel<512, 16, KIND>;
    if (ept <= 4) return mf_fused_kernel<1024, 4, KIND>;
    if (ept <= 8) return mf_fused_kernel<1024, 8, KIND>;
    return mf_fused_kernel<1024, 16, KIND>;
}

static mf_fused_fn mf_fused_for(const gh_ctx *c)
{
    if (c->cell_kind != GH_CELL_TESSEROID) return mf_fused_for_kind<0>(c->mf_T, c->mf_EPT);
    if (!c->mf_near_on) return mf_fused_for_kind<1>(c->mf_T, c->mf_EPT);
    if (c->mf_exact) return mf_fused_for_kind<2>(c->mf_T, c->mf_EPT);
    if (!c->mf_pipe) return mf_fused_for_kind<3>(c->mf_T, c->mf_EPT);
    // the fast pass with the next column's constants and scalars fetched ahead (same bits)
    if (c->mf_T == 256) return c->mf_EPT <= 4 ? mf_tess_fast_kernel<256, 4> : mf_tess_fast_kernel<256, 8>;
    if (c->mf_T == 512) return c->mf_EPT <= 8 ? mf_tess_fast_kernel<512, 8> : mf_tess_fast_kernel<512, 16>;
    if (c->mf_EPT <= 4) return mf_tess_fast_kernel<1024, 4>;
    if (c->mf_EPT <= 8) return mf_tess_fast_kernel<1024, 8>;
    return mf_tess_fast_kernel<1024, 16>;
}

// Tesseroids, fused pass: list of the pairs whose root must be subdivided (or flags an error), with
// their entries.  Kept only while it stays small next to what a stored G would take (1/64 of N*M
// entries and 1/8 of the free memory); otherwise the pass subdivides inside (KIND 1).
static int build_near_table(gh_ctx *c)
{
    c->mf_near_on = false;
    if (env_int("GRAVHMC_MF_NEAR", 1) == 0) return GH_OK;
    // (the shift-invariant store evaluates entries for its table and the compressor's rows only: on a large grid the
    // near-field table -- a pass over all N M pairs -- would cost more than it saves)
    if (c->ls && (double)c->N * (double)c->M > 2e10) return GH_OK;
    const MfGeom g = mf_geom(c);
    int *count = nullptr;
    HIPCHK(c, hipMalloc((void **)&count, sizeof(int) * (size_t)c->M));
    tess_near_count_kernel<<<dim3((unsigned)c->M), dim3(256), 0, c->stream>>>(g, c->mf_cellc, count);
    std::vector<int> hc((size_t)c->M);
    hipError_t e = hipMemcpyAsync(hc.data(), count, sizeof(int) * (size_t)c->M, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    hipFree(count);
    if (e != hipSuccess) return fail(c, GH_ERR_HIP, "near-field count: %s", hipGetErrorString(e));
    std::vector<int64_t> hp((size_t)c->M + 1, 0);
    for (int64_t j = 0; j < c->M; ++j) hp[(size_t)j + 1] = hp[(size_t)j] + hc[(size_t)j];
    const int64_t n = hp[(size_t)c->M];
    size_t free_b = 0, total_b = 0;
    (void)hipMemGetInfo(&free_b, &total_b);
    if (n > c->N * c->M / 64 || (size_t)n * 12 > free_b / 8) return GH_OK;
    TRY(dalloc(c, &c->mf_near_ptr, (size_t)c->M + 1, false));
    TRY(dalloc(c, &c->mf_near_row, (size_t)std::max<int64_t>(n, 1), false));
    TRY(dalloc(c, &c->mf_near_val, (size_t)std::max<int64_t>(n, 1), false));
    HIPCHK(c, hipMemcpyAsync(c->mf_near_ptr, hp.data(), sizeof(int64_t) * ((size_t)c->M + 1), hipMemcpyHostToDevice,
                             c->stream));
    unsigned long long *leaves = nullptr;
    HIPCHK(c, hipMalloc((void **)&leaves, sizeof(unsigned long long)));
    HIPCHK(c, hipMemsetAsync(leaves, 0, sizeof(unsigned long long), c->stream));
    tess_near_fill_kernel<<<dim3((unsigned)c->M), dim3(256), 0, c->stream>>>(g, c->mf_cellc, c->mf_near_ptr,
                                                                           c->mf_near_row, c->mf_near_val, leaves);
    unsigned long long hl = 0;
    e = hipMemcpyAsync(&hl, leaves, sizeof hl, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);  // (hp must outlive the copy above)
    hipFree(leaves);
    if (e != hipSuccess) return fail(c, GH_ERR_HIP, "near-field fill: %s", hipGetErrorString(e));
    c->mf_near_n = n;
    c->mf_near_leaves = (int64_t)hl;
    c->mf_near_on = true;
    return GH_OK;
}

static bool mft_plan(gh_ctx *c);                     // host_batch.h: one chain on teams (mf_team_kernel)
static int mft_launch(gh_ctx *c, SweepArgs &a);
static int launch_lonsym(gh_ctx *c, SweepArgs &a);  // host_lonsym.h
static bool shard_rows(const gh_ctx *c);                            // host_comm.h
static int comm_allreduce(gh_ctx *c, double *buf, int64_t count);
static bool lonsym_on(const gh_ctx *c);
static bool lonsym_harmonic(const gh_ctx *c);
static bool lonsym_one_row(const gh_ctx *c);
static int lonsym_post_now(gh_ctx *c);
static int lonsym_epilogue_check(gh_ctx *c);
static int lonsym_classes(const gh_ctx *c);
static int lonsym_grid(const gh_ctx *c);
static int64_t lonsym_table_bytes(const gh_ctx *c);

// Partition of the matrix-free passes.  N <= 16384: the fused pass (one workgroup per column at a
// time, columns dealt round-robin); else the two-pass form (one wave per cell for the adjoint,
// chunks of cells per forward partial).
static int configure_mf(gh_ctx *c)
{
    if (lonsym_on(c)) {
        c->grid = lonsym_grid(c);  // one workgroup per cell row at a time, rows dealt round-robin
        c->n_teams = c->grid;
        return GH_OK;
    }
    if (c->mf_fused) {
        const int64_t ld = c->ld;
        c->mf_T = ld <= 2048 ? 256 : 1024;
        {
            // diagnostic override of the workgroup size (must still hold a column: 8 / 16 / 16 rows per thread)
            const int t_env = env_int("GRAVHMC_MF_T", 0);
            if ((t_env == 256 && ld <= 2048) || (t_env == 512 && ld <= 8192) || t_env == 1024) c->mf_T = t_env;
        }
        const int e = (int)((ld + c->mf_T - 1) / c->mf_T);
        c->mf_EPT = c->mf_T == 256 ? (e <= 4 ? 4 : 8) : c->mf_T == 512 ? (e <= 8 ? 8 : 16) : (e <= 4 ? 4 : e <= 8 ? 8 : 16);
        c->mf_lds = ((size_t)c->mf_T * c->mf_EPT + 2 * (c->mf_T / 64 + 8)) * sizeof(double);
        {
            // the pipelined tesseroid pass also keeps r, the parked constants and scalars in LDS
            const size_t pipe_lds = (2 * (size_t)c->mf_T * c->mf_EPT + 2 * (c->mf_T / 64 + 8) + 64 + 48) * sizeof(double);
            c->mf_pipe = c->cell_kind == GH_CELL_TESSEROID && c->mf_near_on && !c->mf_exact &&
                         env_int("GRAVHMC_MF_PIPE", 1) != 0 && pipe_lds <= 160 * 1024;
            if (c->mf_pipe) c->mf_lds = pipe_lds;
        }
        mf_fused_fn f = mf_fused_for(c);
        HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(f), c->mf_lds));
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reinterpret_cast<const void *>(f), c->mf_T,
                                                         c->mf_lds) != hipSuccess || occ < 1) {
            (void)hipGetLastError();
            occ = 1;
        }
        occ = env_int("GRAVHMC_MF_WG_PER_CU", occ);
        c->grid = (int)std::min<int64_t>(c->M, (int64_t)c->cus * occ);
        c->n_teams = c->grid;  // one partial of p^2 per workgroup
        return GH_OK;
    }
    c->n_teams = (int)((c->M + 3) / 4);
    const int64_t chunks = std::min<int64_t>(c->M, std::max<int64_t>(1, (int64_t)c->cus * 16 / std::max<int64_t>(1, (c->ld + 255) / 256)));
    c->mf_cells_per_chunk = (c->M + chunks - 1) / chunks;
    c->grid = (int)((c->M + c->mf_cells_per_chunk - 1) / c->mf_cells_per_chunk);
    return GH_OK;
}

// matrix-free counterpart of one sweep: the fused pass, or adjoint/update pass then forward pass
static int launch_mf(gh_ctx *c, SweepArgs &a)
{
    const MfGeom g = mf_geom(c);
    const double *wm = c->weighted ? c->wm : nullptr;
    bool timed = c->prof && c->ev_used + 2 <= c->ev.size();
    if ((a.mode & SW_ADJ) && !wm) return fail(c, GH_ERR_ARG, "matrix-free adjoint needs gh_weight first");
    if (timed) HIPCHK(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    if (lonsym_on(c)) {
        TRY(launch_lonsym(c, a));
    } else if ((a.mode & SW_ADJ) && (a.mode & (SW_UPD | SW_PFIN)) && c->chain_teams_ok && mft_plan(c)) {
        // a leapfrog step of the chain: teams of workgroups (the trajectory code looks at the abort word)
        TRY(mft_launch(c, a));
    } else if (c->mf_fused) {
        a.ld = c->ld;
        a.M = c->M;
        MfNear near{c->mf_near_ptr, c->mf_near_row, c->mf_near_val};
        hipLaunchKernelGGL(mf_fused_for(c), dim3(c->grid), dim3(c->mf_T), c->mf_lds, c->stream, g, a,
                           wm, c->cell_kind == GH_CELL_TESSEROID ? c->mf_cellc : nullptr, near,
                           c->prof ? c->mf_stats : nullptr);
    } else {
        if (a.mode & SW_ADJ)
            mf_adjoint_kernel<<<dim3((unsigned)((c->M + 3) / 4)), dim3(256), 0, c->stream>>>(g, a, wm);
        if (a.mode & SW_FWD) {
            const double *x = (a.mode & SW_UPD) ? a.x_out : a.x_in;
            mf_forward_kernel<<<dim3((unsigned)((c->ld + 255) / 256), (unsigned)c->grid), dim3(256), 0,
                                c->stream>>>(g, x, wm, c->mf_cells_per_chunk, c->ld, a.slab);
        }
    }
    if (timed) {
        HIPCHK(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        c->ev_bytes[c->ev_used / 2] = lonsym_on(c) ? lonsym_table_bytes(c) : 0;  // (the table is read once per pass)
        c->ev_used += 2;
    }
    if (c->prof) c->mf_launches += 1;
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

static int launch_sweep_one(gh_ctx *c, SweepArgs &a)
{
    a.G = c->G;
    a.ld = c->ld;
    a.M = c->M;
    a.cols_per_team = c->cols_per_team;
    a.n_teams = c->n_teams_sweep;
    const int threads = (c->TW == 1 ? 4 : c->TW) * 64;
    sweep_fn f = sweep_for(c);
    // short sweeps: an event pair costs about as much as the kernel, time every 16th launch only
    bool timed = c->prof && c->ev_used + 2 <= c->ev.size() && (c->prof_seen++ % c->prof_stride) == 0;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    hipLaunchKernelGGL(f, dim3(c->grid), dim3(threads), c->lds_bytes, c->stream, a);
    if (timed) {
        HIPCHK(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        c->ev_bytes[c->ev_used / 2] = a.rows < c->ld ? a.rows * c->M * (int64_t)sizeof(double)
                                                      : c->N * c->M * (int64_t)sizeof(double);
        c->ev_used += 2;
    }
    if (c->prof) c->prof_launches += 1;
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

// ---- team sweep: one read of G per fused step for N > 16384 (teamsweep.hip.h)

typedef void (*team_fn)(TeamArgs);

// Instantiations: 1024 threads x 5 double2 x 3 register buffers, with a lag of one or two columns
// (teamsweep.hip.h).  C5 share (N = 4*10^4, 3*10^5 cells, 96 GB), ms per team sweep: lag 2 14.2,
// lag 1 14.7 (a plain read of the matrix: 14.0).  Other shapes were measured only while the kernel's
// column requests were still drained in front of every dot (18.0 then): 1024 x 4 x 4 (5 members, 240
// of 256 CUs) 19.8; 512 x 10 x 4 17.5 + more variance; 512 x 10 x 5 25.8; 512 x 8 x 5 21.4.
static team_fn team_kernel_for(int threads, int ept2, int depth, int lag)
{
    if (threads == 1024 && ept2 == 5 && depth == 3)
        return lag == 2 ? teamsweep_kernel<1024, 5, 3, true> : teamsweep_kernel<1024, 5, 3, false>;
    return nullptr;
}

static bool team_plan(gh_ctx *c)
{
    gh_ctx::Team &t = c->tm;
    if (t.state != 0) return t.state > 0;
    t.state = -1;
    if (env_int("GRAVHMC_TEAM", 1) == 0) return false;
    if (c->mf || c->n_panels < 2 || !c->G) return false;
    // instantiation: threads x double2 per thread x column buffers
    t.threads = 1024;
    t.ept2 = 5;
    t.depth = 3;
    t.lag = env_int("GRAVHMC_TEAM_LAG", 2) == 1 ? 1 : 2;
    const size_t lds_small = (2 * TS_MAXWAVES + 4) * sizeof(double);
    int64_t cap = (int64_t)t.threads * t.ept2 * 2;  // rows a member holds of a column
    // lag 2: the member's part of r and the parked column share the 160 KB of LDS
    if (t.lag == 2) cap = std::min<int64_t>(cap, (int64_t)((160 * 1024 - lds_small) / (2 * sizeof(double))) / 16 * 16);
    t.Q = (int)((c->ld + cap - 1) / cap);
    if (t.Q < 2) t.Q = 2;
    if (t.Q > TS_MAXQ || c->cus < 8 * t.Q) return false;
    t.panel_rows = ((c->ld + t.Q - 1) / t.Q + 15) / 16 * 16;
    t.tpx = std::min(32, c->cus / 8) / t.Q;  // blocks with equal blockIdx % 8 share an XCD: 32 CUs each
    if (t.tpx < 1) return false;
    t.grid = 8 * t.tpx * t.Q;
    const int n_teams = 8 * t.tpx;
    t.cols_per_team = (c->M + n_teams - 1) / n_teams;
    t.lds = (size_t)t.panel_rows * (size_t)t.lag * sizeof(double) + lds_small;
    team_fn f = team_kernel_for(t.threads, t.ept2, t.depth, t.lag);
    if (allow_dynamic_lds(reinterpret_cast<const void *>(f), t.lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(f), t.threads,
                                                     t.lds) != hipSuccess ||
        per_cu < 1 || (int64_t)per_cu * c->cus < t.grid) {
        (void)hipGetLastError();
        return false;
    }
    const size_t ng = (size_t)n_teams * TS_MAXQ * TS_RING * 2;
    if (dalloc(c, &t.gran, ng) != GH_OK || dalloc(c, &t.abort_w, 4) != GH_OK) return false;
    t.tag = 0;
    t.state = 1;
    return true;
}

static int launch_team(gh_ctx *c, const SweepArgs &full)
{
    gh_ctx::Team &t = c->tm;
    if ((uint64_t)t.tag + (uint64_t)t.cols_per_team + 2 > 0xf0000000ull) {
        // 32-bit tags about to wrap: start again on zeroed granules
        HIPCHK(c, hipMemsetAsync(t.gran, 0, (size_t)8 * t.tpx * TS_MAXQ * TS_RING * 2 * sizeof(u64), c->stream));
        t.tag = 0;
    }
    TeamArgs a{};
    a.s = full;
    a.s.G = c->G;
    a.s.ld = c->ld;
    a.s.M = c->M;
    a.Q = t.Q;
    // test hook: the members wait for a part that never comes, time out and give up
    a.poll_q = t.Q + ((env_int("GRAVHMC_TEAM_TEST_ABORT", 0) && t.Q < TS_MAXQ) ? 1 : 0);
    a.tpx = t.tpx;
    a.panel_rows = t.panel_rows;
    a.cols_per_team = t.cols_per_team;
    a.n_pp = c->n_teams;
    a.gran = t.gran;
    a.tag0 = t.tag;
    a.abort_w = t.abort_w;
    bool timed = c->prof && c->ev_used + 2 <= c->ev.size() && (c->prof_seen++ % c->prof_stride) == 0;
    if (timed) HIPCHK(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    hipLaunchKernelGGL(team_kernel_for(t.threads, t.ept2, t.depth, t.lag), dim3(t.grid), dim3(t.threads), t.lds, c->stream, a);
    if (timed) {
        HIPCHK(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        c->ev_bytes[c->ev_used / 2] = c->N * c->M * (int64_t)sizeof(double);
        c->ev_used += 2;
    }
    if (c->prof) c->prof_launches += 1;
    HIPCHK(c, hipGetLastError());
    t.tag += (unsigned)t.cols_per_team + 1u;
    t.inflight = true;
    t.launches += 1;
    return GH_OK;
}

// Bookkeeping of a team sweep that gave up.  On a sharded chain EVERY rank calls this when ANY rank's
// sweep gave up (the decision travels with the trajectory's scalar all-reduce): the garbage of the
// failed rank's slab has been summed into everybody's d and r, and the ranks must repeat the
// trajectory together, with the same number of collectives and the same count of time-outs.
static int team_mark_failed(gh_ctx *c, const char *whose)
{
    gh_ctx::Team &t = c->tm;
    if (t.state == -1 || !t.gran) return GH_OK;  // (teams not in use on this rank)
    t.aborts += 1;
    const bool for_good = t.aborts >= 3;
    fprintf(stderr, "libgravhmc: %s team sweep timed out waiting for its workgroups (%d of 3); repeating %s in row panels\n",
            whose, t.aborts, for_good ? "this and everything after it" : "the step");
    HIPCHK(c, hipMemsetAsync(t.abort_w, 0, 4 * sizeof(unsigned), c->stream));
    HIPCHK(c, hipMemsetAsync(t.gran, 0, (size_t)8 * t.tpx * TS_MAXQ * TS_RING * 2 * sizeof(u64), c->stream));
    t.tag = 0;
    t.state = for_good ? -1 : 2;  // 2: skip the teams until the repeated work is done (team_resume)
    return GH_OK;
}

// After a synchronisation point: did a team sweep since the last look give up (its workgroups were
// not all resident)?  Then everything it fed is void: the caller repeats its work, which now runs in
// row panels (again on teams after a transient stall; for good after three).
static int team_failed(gh_ctx *c, bool *failed)
{
    gh_ctx::Team &t = c->tm;
    *failed = false;
    if (!t.inflight) return GH_OK;
    t.inflight = false;
    unsigned w[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(w, t.abort_w, sizeof w, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    t.late_polls = w[1];  // columns whose parts some member had to wait for (all launches so far)
    if (w[0] == 0u) return GH_OK;
    *failed = true;
    return team_mark_failed(c, "this rank's");
}

static void team_resume(gh_ctx *c)
{
    if (c->tm.state == 2) c->tm.state = 1;
}

static int launch_sweep(gh_ctx *c, SweepArgs &a)
{
    if (a.mode & SW_FWD) {
        c->slab_live = c->grid;
        c->dsum_live = false;
        c->dsum_n = 0;
    }
    if (c->mf) {
        if (lonsym_on(c) && (a.mode & SW_FWD) && c->dsum) {
            a.dsum = c->dsum;  // (the shift-invariant pass delivers the sums of its slab rows as well)
            c->dsum_live = true;
        }
        if (lonsym_one_row(c) && (a.mode & SW_FWD)) {
            // (harmonic form: ONE finished slab row; the sums come per class of observations)
            c->slab_live = 1;
            c->dsum_n = lonsym_classes(c);
        }
        return launch_mf(c, a);
    }
    if (shard_rows(c)) {
        // Row blocks: the dot of a column with r spans the ranks.  Adjoint of the local rows (panel by panel)
        // into the gradient buffer -- alpha grad R added by rank 0 only --, all-reduce of its M doubles,
        // elementwise update (replicated: every rank holds the whole model), forward of the local rows: TWO
        // reads of the local shard per leapfrog step.
        const SweepArgs full = a;
        auto panel = [&](SweepArgs &s, int p) {
            s.row0 = (int64_t)p * c->panel_rows;
            s.rows = c->n_panels == 1 ? c->ld : std::min<int64_t>(c->panel_rows, c->ld - s.row0);
        };
        if (full.mode & SW_ADJ) {
            double *gdst = (full.mode & SW_GOUT) ? full.g_out : c->gbuf;
            for (int p = 0; p < c->n_panels; ++p) {
                SweepArgs s = full;
                s.mode = SW_ADJ | SW_GOUT | (p ? SW_GACC : 0);
                s.greg = (p == 0 && c->sh.rank == 0) ? full.greg : nullptr;
                s.g_out = gdst;
                panel(s, p);
                TRY(launch_sweep_one(c, s));
            }
            TRY(comm_allreduce(c, gdst, c->M));
            if (full.mode & (SW_UPD | SW_PFIN)) {
                SweepArgs u = full;
                vec_update_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(u, gdst, c->M, c->n_teams);
                HIPCHK(c, hipGetLastError());
            }
        }
        if (full.mode & SW_FWD) {
            for (int p = 0; p < c->n_panels; ++p) {
                SweepArgs s = full;
                s.mode = SW_FWD;
                s.x_in = (full.mode & SW_UPD) ? full.x_out : full.x_in;
                panel(s, p);
                TRY(launch_sweep_one(c, s));
            }
        }
        return GH_OK;
    }
    if (c->n_panels == 1) {
        a.row0 = 0;
        a.rows = c->ld;
        if ((a.mode & SW_FWD) && c->TW > 1 && c->dsum) {
            a.dsum = c->dsum;  // sums of the slab rows: the epilogue then needs one launch
            c->dsum_live = true;
        }
        return launch_sweep_one(c, a);
    }
    // more rows than one workgroup holds of a column.  The fused step (adjoint + update + forward)
    // runs on teams of workgroups: one read of G.  Adjoint-only and forward-only sweeps read G once
    // in row panels anyway.
    if ((a.mode & SW_ADJ) && (a.mode & SW_FWD) && !(a.mode & SW_GACC) && team_plan(c) && c->tm.state == 1) {
        c->slab_live = 8 * c->tm.tpx;
        return launch_team(c, a);
    }
    // row panels: adjoint of every panel into gbuf, elementwise update, forward of every panel
    const SweepArgs full = a;
    if (full.mode & SW_ADJ) {
        double *gdst = (full.mode & SW_GOUT) ? full.g_out : c->gbuf;
        for (int p = 0; p < c->n_panels; ++p) {
            SweepArgs s = full;
            s.mode = SW_ADJ | SW_GOUT | (p ? SW_GACC : 0);
            s.greg = p ? nullptr : full.greg;
            s.g_out = gdst;
            s.row0 = (int64_t)p * c->panel_rows;
            s.rows = std::min<int64_t>(c->panel_rows, c->ld - s.row0);
            TRY(launch_sweep_one(c, s));
        }
        if (full.mode & (SW_UPD | SW_PFIN)) {
            SweepArgs u = full;
            vec_update_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(u, gdst, c->M, c->n_teams);
            HIPCHK(c, hipGetLastError());
        }
    }
    if (full.mode & SW_FWD) {
        for (int p = 0; p < c->n_panels; ++p) {
            SweepArgs s = full;
            s.mode = SW_FWD;
            s.x_in = (full.mode & SW_UPD) ? full.x_out : full.x_in;
            s.row0 = (int64_t)p * c->panel_rows;
            s.rows = std::min<int64_t>(c->panel_rows, c->ld - s.row0);
            TRY(launch_sweep_one(c, s));
        }
    }
    return GH_OK;
}
