// libgravhmc host side: planning and launching the persistent harmonic pass of the shift-invariant store
// (lonres.hip.h).  Included once by gravhmc.hip.
#pragma once

typedef void (*lonres_fn_t)(LonResArgs);
static lonres_fn_t lonres_fn(int rw)
{
    return rw <= 1 ? lonsymh_resident_kernel<1> : rw == 2 ? lonsymh_resident_kernel<2> : rw == 3 ? lonsymh_resident_kernel<3>
                                                                                               : lonsymh_resident_kernel<4>;
}

// Can the chain of this context run inside lonsymh_resident_kernel?  The harmonic store on one device, every
// workgroup of the pass resident (one per CU: its rows of the table live in its registers) and all cell rows in
// ONE group of rows per workgroup, a class owner for every class.
static bool lonres_plan(gh_ctx *c)
{
    if (!lonsym_harmonic(c)) return false;
    LonSymHost &h = *c->ls;
    LonSymHost::Res &r = h.res;
    if (r.state != 0) return r.state > 0;
    r.state = -1;
    if (env_int("GRAVHMC_LONSYM_RESIDENT", 1) == 0) return false;
    if (c->sh.kind != 0 || c->wv.on) return false;
    if (h.hgrid > LR_MAXWG || h.hgrid < h.na || (int64_t)h.hgrid * h.rw < h.nc || h.na > 64 || h.n > 126 || h.n < 2 || h.max_extra > 2) return false;
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess) return false;
    r.lds = lonres_lds_doubles(h.n, h.nf, h.na, h.rw) * sizeof(double);
    if (r.lds > (size_t)lds_max) return false;
    lonres_fn_t f = lonres_fn(h.rw);
    if (allow_dynamic_lds(reinterpret_cast<const void *>(f), r.lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(f), LR_THREADS, r.lds) != hipSuccess ||
        per_cu < 1 || (int64_t)per_cu * c->cus < h.hgrid) {
        (void)hipGetLastError();
        return false;
    }
    r.state = 1;
    return true;
}

// lonres_plan + what depends on the regulariser set at the moment (the stencil kinds need the model's shape)
static bool lonres_usable(gh_ctx *c)
{
    if (!lonres_plan(c)) return false;
    return c->reg_kind >= 0 && c->reg_kind <= 3 && (c->reg_kind == 0 || c->reg_kind == 2 || c->shape[0] > 0);
}

// K trajectories of the context's chain in one launch (same contract as gh_chain_run / chain_run_resident).
// GH_RESIDENT_ABORTED: the kernel gave up waiting for its workgroups, nothing was changed.
// st: the context whose chain runs (c itself, or one of the light contexts of a batch of chains: the launch, its
// tables and exchange buffers are c's).  p0rows (or nullptr): the K momentum rows where they lie, instead of p0s.
static int chain_run_lonres(gh_ctx *c, gh_ctx *st, int K, const int *L, const double *p0s, const double *const *p0rows, const double *us,
                            double dt, int64_t stop_at_accepts, int64_t record_from, int *accepted, double *out5s, double *x_out,
                            int *n_run)
{
    LonSymHost &h = *c->ls;
    LonSymHost::Res &r = h.res;
    const size_t M = (size_t)c->M;
    const int nwg = h.hgrid;
    const size_t E = (size_t)h.na * (size_t)h.nf;
    HIPCHK(c, hipSetDevice(c->device));
    if (!r.slab) {
        TRY(dalloc(c, &r.slab, ((size_t)nwg + 8) * E));
        TRY(dalloc(c, &r.flagg, (size_t)nwg + 8));
        TRY(dalloc(c, &r.xccg, (size_t)nwg + 8));
        TRY(dalloc(c, &r.xslabg, 2 * (size_t)RES_CLUSTERS * E * 2));
        TRY(dalloc(c, &r.rhatg, 2 * E));
        TRY(dalloc(c, &r.clsg, 2 * 64 * 4));
        TRY(dalloc(c, &r.scalg, 2 * (size_t)LR_MAXWG * 2));
        TRY(dalloc(c, &r.ppg, 2 * (size_t)LR_MAXWG * 2));
        TRY(dalloc(c, &r.xpub, 2 * (size_t)c->M));
        TRY(dalloc(c, &r.abort_w, 4));
        TRY(dalloc(c, &r.n_run, 4));
        TRY(dalloc(c, &r.ucur, 4));
        // M^ = the transform of the slots' observation counts: R^ of a residual of ones
        TRY(dalloc(c, &r.mhat, E));
        {
            double *ones = nullptr;
            TRY(dalloc(c, &ones, (size_t)c->ld));
            const std::vector<double> hv((size_t)c->N, 1.0);
            HIPCHK(c, hipMemcpyAsync(ones, hv.data(), hv.size() * sizeof(double), hipMemcpyHostToDevice, c->stream));
            lonsymh_rhat_kernel<<<dim3((unsigned)h.na), dim3(256), 0, c->stream>>>(lonsymh_geom(c), ones);
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipMemcpyAsync(r.mhat, h.Rhat, E * sizeof(ghk::d2), hipMemcpyDeviceToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            h.rhat_of = nullptr;
        }
        if (env_int("GRAVHMC_LONSYM_TIMING", 0)) TRY(dalloc(c, &r.dbg, 16));
        HIPCHK(c, hipEventCreate(&r.ev0));
        HIPCHK(c, hipEventCreate(&r.ev1));
    }
    const bool want_x = x_out != nullptr || st->ring != nullptr;
    if (K > r.Kcap) {
        const int cap = std::max(K, 32);
        r.L = r.accepted = nullptr;
        r.p0s = r.us = r.out5s = r.xacc = nullptr;
        TRY(dalloc(c, &r.L, (size_t)cap));
        TRY(dalloc(c, &r.accepted, (size_t)cap));
        TRY(dalloc(c, &r.p0s, (size_t)cap * M, false));
        TRY(dalloc(c, &r.us, (size_t)cap));
        TRY(dalloc(c, &r.out5s, (size_t)cap * 5));
        TRY(dalloc(c, &r.xacc, (size_t)cap * M, false));
        r.Kcap = cap;
    }
    int64_t steps = 1;
    for (int k = 0; k < K; ++k) steps += L[k];
    if (r.dirty || (uint64_t)r.tag + (uint64_t)steps + 2 > 0xf0000000ull || (uint64_t)r.tagE + (uint64_t)K + 2 > 0xf0000000ull ||
        r.ltag > 0xf0000000u) {
        HIPCHK(c, hipMemsetAsync(r.flagg, 0, ((size_t)nwg + 8) * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.xccg, 0, ((size_t)nwg + 8) * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.xslabg, 0, 2 * (size_t)RES_CLUSTERS * E * 2 * sizeof(ghk::u32x4), c->stream));
        HIPCHK(c, hipMemsetAsync(r.clsg, 0, 2 * 64 * 4 * sizeof(ghk::u32x4), c->stream));
        HIPCHK(c, hipMemsetAsync(r.scalg, 0, 2 * (size_t)LR_MAXWG * 2 * sizeof(ghk::u32x4), c->stream));
        HIPCHK(c, hipMemsetAsync(r.ppg, 0, 2 * (size_t)LR_MAXWG * 2 * sizeof(ghk::u32x4), c->stream));
        r.tag = r.tagE = r.ltag = 0;
        r.dirty = false;
    }
    HIPCHK(c, hipMemsetAsync(r.abort_w, 0, 4 * sizeof(unsigned), c->stream));
    // the momenta: one pinned staging buffer, one copy (the caller's array is pageable)
    if ((size_t)K * M > r.h_stage_n) {
        if (r.h_stage) HIPCHK(c, hipHostFree(r.h_stage));
        r.h_stage = nullptr;
        r.h_stage_n = (size_t)std::max(K, 32) * M;
        HIPCHK(c, hipHostMalloc((void **)&r.h_stage, r.h_stage_n * sizeof(double)));
    }
    {
        // rows inside a block of gh_pinned_alloc go straight from where they lie, adjacent ones in one copy; the others
        // are gathered into the pinned staging buffer first (host_resbatch.h)
        std::vector<const double *> src((size_t)K);
        std::vector<char> direct((size_t)K, 0);
        int n_staged = 0;
        for (int k = 0; k < K; ++k) {
            src[(size_t)k] = p0rows ? p0rows[(size_t)k] : p0s + (size_t)k * M;
            const char *lo = (const char *)src[(size_t)k], *hi = lo + M * sizeof(double);
            for (const gh_ctx::Pinned &pm : c->pinned)
                if (lo >= pm.base && hi <= pm.base + pm.bytes) {
                    direct[(size_t)k] = 1;
                    break;
                }
            n_staged += direct[(size_t)k] ? 0 : 1;
        }
        if (n_staged > 0) {
            auto part = [&](int k0, int k1) {
                for (int k = k0; k < k1; ++k)
                    if (!direct[(size_t)k]) memcpy(r.h_stage + (size_t)k * M, src[(size_t)k], M * sizeof(double));
            };
            const int nthr = (size_t)n_staged * M * sizeof(double) >= ((size_t)1 << 20) ? std::min(4, K) : 1;
            std::vector<std::thread> pool;
            for (int i = 1; i < nthr; ++i) pool.emplace_back(part, (int)((int64_t)K * i / nthr), (int)((int64_t)K * (i + 1) / nthr));
            part(0, K / nthr);
            for (std::thread &th : pool) th.join();
        }
        for (int k0 = 0; k0 < K;) {
            int k1 = k0 + 1;
            const double *from = direct[(size_t)k0] ? src[(size_t)k0] : r.h_stage + (size_t)k0 * M;
            while (k1 < K && direct[(size_t)k1] == direct[(size_t)k0] && (direct[(size_t)k0] ? src[(size_t)k1] == src[(size_t)k1 - 1] + M : true))
                ++k1;
            HIPCHK(c, hipMemcpyAsync(r.p0s + (size_t)k0 * M, from, (size_t)(k1 - k0) * M * sizeof(double), hipMemcpyHostToDevice, c->stream));
            k0 = k1;
        }
        HIPCHK(c, hipMemcpyAsync(r.us, us, (size_t)K * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(r.L, L, (size_t)K * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    LonResArgs a{};
    a.g = lonsymh_geom(c);
    a.g.dbg = nullptr;
    a.N = c->N;
    a.M = c->M;
    a.nwg = nwg;
    if (env_int("GRAVHMC_LONRES_TEST_ABORT", 0)) a.nwg += 8;  // test hook: partners that do not exist
    a.try_local = env_int("GRAVHMC_RESIDENT_LOCAL", 1);
    a.wm = st->weighted ? st->wm : nullptr;
    a.low = st->low;
    a.high = st->high;
    a.mwapr = st->mwapr;
    a.wm2 = st->wm2;
    a.kind = st->reg_kind;
    a.nz = st->shape[0];
    a.ny = st->shape[1];
    a.nx = st->shape[2];
    a.xpub = r.xpub;
    a.ms_grad_den_mw = 0;
    a.alpha = st->alpha;
    a.beta = st->beta;
    a.dobs_c = st->dobs_c;
    a.gfix = st->have_fix ? st->gfix : nullptr;
    a.gfix_sum = st->have_fix ? st->gfix_sum : 0.0;
    a.Mhat = r.mhat;
    a.x_cur = st->xb[st->xcur];
    a.K = K;
    a.L = r.L;
    a.p0s = r.p0s;
    a.us = r.us;
    a.dt = dt;
    a.stop_at_accepts = stop_at_accepts;
    a.accept_count0 = st->accept_count;
    a.accepted = r.accepted;
    a.out5s = r.out5s;
    a.xacc = want_x ? r.xacc : nullptr;
    a.n_run = r.n_run;
    a.ucur = r.ucur;
    a.slab = r.slab;
    a.flagg = r.flagg;
    a.xslabg = r.xslabg;
    a.rhatg = r.rhatg;
    a.clsg = r.clsg;
    a.scalg = r.scalg;
    a.ppg = r.ppg;
    a.xccg = r.xccg;
    a.tag0 = r.tag;
    a.tagE0 = r.tagE;
    a.ltag = r.ltag + 1u;
    a.abort_w = r.abort_w;
    a.dbg = r.dbg;
    lonres_fn_t f = lonres_fn(h.rw);
    HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(f), r.lds));
    if (c->prof) HIPCHK(c, hipEventRecord(r.ev0, c->stream));
    // (a plain launch: the grid was checked against the occupancy query in lonres_plan; every wait inside is bounded)
    hipLaunchKernelGGL(f, dim3((unsigned)nwg), dim3(LR_THREADS), r.lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    if (c->prof) HIPCHK(c, hipEventRecord(r.ev1, c->stream));
    unsigned h_sync[4] = {0, 0, 0, 0};
    int h_run[4] = {0, 0, 0, 0};
    double h_u[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(h_sync, r.abort_w, sizeof h_sync, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_run, r.n_run, sizeof h_run, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_u, r.ucur, 3 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(accepted, r.accepted, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out5s, r.out5s, (size_t)K * 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    r.ltag += 1u;
    if (h_sync[0] != 0u) {
        r.aborts += 1;
        r.dirty = true;
        const bool for_good = r.aborts >= 3;
        if (for_good) r.state = -1;
        fprintf(stderr, "libgravhmc: the persistent harmonic pass timed out waiting for its workgroups (%d of 3); %s with one "
                        "launch per phase\n", r.aborts, for_good ? "continuing for good" : "running this batch");
        return GH_RESIDENT_ABORTED;
    }
    r.tag += (unsigned)h_run[1];
    r.tagE += (unsigned)h_run[2];
    r.launches += 1;
    r.evals += h_run[1];
    r.trajectories += h_run[0];
    if (c->prof) {
        float t = 0.f;
        HIPCHK(c, hipEventElapsedTime(&t, r.ev0, r.ev1));
        c->prof_ms_acc += t;
        c->prof_res_evals += h_run[1];
    }
    *n_run = h_run[0];
    for (int k = 0; k < h_run[0]; ++k) {
        if (!accepted[k]) continue;
        st->accept_count += 1;
        if (st->ring && st->accept_count > record_from) {
            ring_store_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(
                r.xacc + (size_t)k * M, c->weighted ? c->wm : nullptr, c->M, st->ring + (size_t)st->ring_next * M);
            st->ring_next = (st->ring_next + 1) % st->ring_K;
            st->ring_count += 1;
        }
        if (x_out)
            HIPCHK(c, hipMemcpyAsync(x_out + (size_t)k * M, r.xacc + (size_t)k * M, M * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // (d, r and the scalars of the current sample are behind x now: chain_state_fresh brings them up to date for
    // whoever reads them next)
    st->U_cur[0] = h_u[0];
    st->U_cur[1] = h_u[1];
    st->U_cur[2] = h_u[2];
    st->spec_valid = st->pn_valid = false;
    st->st_stale = true;
    return GH_OK;
}
