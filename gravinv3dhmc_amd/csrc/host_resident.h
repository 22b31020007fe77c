// libgravhmc host side: planning and launching the resident chain kernel (resident.hip.h).
// Included once by gravhmc.hip.
#pragma once

typedef void (*resident_fn)(ResArgs);
enum { GH_RESIDENT_ABORTED = 1000 };  // internal: chain_run_resident gave up, state untouched

// rc: double2 chunks per lane and column; cw: columns a wave keeps in registers (0: none, dots
// read LDS).  Only register copies of at most 20 double2 (80 VGPRs: no spills) are compiled.
template <int RC>
static resident_fn resident_for_rc(int cw)
{
    switch (cw) {
    case 1: return resident_chain_kernel<RC, 1>;
    case 2: if constexpr (RC * 2 <= 20) return resident_chain_kernel<RC, 2>; break;
    case 3: if constexpr (RC * 3 <= 20) return resident_chain_kernel<RC, 3>; break;
    case 4: if constexpr (RC * 4 <= 20) return resident_chain_kernel<RC, 4>; break;
    }
    return resident_chain_kernel<RC, 0>;
}

static resident_fn resident_for(int rc, int cw)
{
    switch (rc) {
    case 1: return resident_for_rc<1>(cw);
    case 2: return resident_for_rc<2>(cw);
    case 3: return resident_for_rc<3>(cw);
    case 4: return resident_for_rc<4>(cw);
    case 5: return resident_for_rc<5>(cw);
    case 6: return resident_for_rc<6>(cw);
    case 7: return resident_for_rc<7>(cw);
    case 8: return resident_for_rc<8>(cw);
    }
    return nullptr;
}

// Can this problem run on the resident chain kernel?  Dense stored G on one device, N <= 1024,
// and one column block per CU that fits the CU's LDS next to the kernel's scratch.
static bool resident_plan(gh_ctx *c)
{
    gh_ctx::Resident &r = c->rs;
    if (r.state != 0) return r.state > 0;
    r.state = -1;
    if (env_int("GRAVHMC_RESIDENT", 1) == 0) return false;
    if (c->mf || c->sh.kind != 0 || c->n_panels != 1 || c->ld > 1024 || !c->G) return false;
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess)
        return false;
    // (GRAVHMC_RESIDENT_WGS: diagnostic -- fewer, larger workgroups)
    const int wgs = std::max(1, std::min(c->cus, env_int("GRAVHMC_RESIDENT_WGS", c->cus)));
    const int cpw = (int)((c->M + wgs - 1) / wgs);
    if (cpw > RES_THREADS) return false;
    r.lds_max = lds_max;
    r.cpw = cpw;
    r.nwg = (int)((c->M + cpw - 1) / cpw);
    if (r.nwg > RES_MAX_WG) return false;
    r.rc = (int)((c->ld / 2 + 63) / 64);
    const bool regs = env_int("GRAVHMC_RESIDENT_REGS", 1) != 0;
    size_t lds = resident_lds_doubles(c->ld, cpw, 1, cpw, false) * sizeof(double);
    r.split = false;
    r.stream = false;
    r.lds_cols = cpw;
    if (lds <= (size_t)lds_max) {
        // every column in LDS; where the registers allow it (what resident_for compiles), the
        // waves keep a second copy of their columns for the dots pass
        r.ct = (regs && cpw <= 4 * RES_WAVES) ? (cpw + RES_WAVES - 1) / RES_WAVES : 0;
        if (r.ct * r.rc > 20) r.ct = 0;
        // wavelet-compressed forward: LDS holds its dense model-space form, the dots need their own
        // (register) copy of Aw
        if (c->wv.on && (r.ct == 0 || wavelet_dense_form(c) != GH_OK)) return false;
    } else {
        // too large for the LDS alone: ONE copy, the first 8 ct columns of a workgroup with the waves
        // (registers: up to 20 double2 per lane), the rest in LDS next to 8 x ld doubles of scratch.
        // As few register columns as the LDS allows: their forward share goes through per-wave
        // partials and an LDS sum (measured at 625 x 10400: ct 3 -> 7.9, ct 4 -> 8.4 us per evaluation).
        const int ct_max = std::min(4, 20 / r.rc);
        const int ct_env = env_int("GRAVHMC_RESIDENT_CT", 0);  // diagnostic: force the number
        r.split = true;
        r.ct = 0;
        for (int ct = 1; ct <= ct_max; ++ct) {
            if (ct_env > 0 && ct != std::min(ct_env, ct_max)) continue;
            const int lc = std::max(0, cpw - ct * RES_WAVES);
            lds = resident_lds_doubles(c->ld, cpw, 1, lc, true) * sizeof(double);
            if (lds <= (size_t)lds_max) {
                r.ct = ct;
                r.lds_cols = lc;
                break;
            }
        }
        if (!regs || c->wv.on) r.ct = 0;  // (one-copy mode needs the register copies and Gl == G)
        if (r.ct < 1) {
            // Larger than the chip holds: as many columns per workgroup as fit stay in LDS, the rest
            // is read from L2 / Infinity Cache in both passes of every evaluation.  Still one launch
            // per batch of trajectories and ~3 us of exchange per evaluation instead of 3-11
            // launches per step; worth it while the streamed bytes (two passes; with the wavelet
            // forward: Aw for the dots and the dense compressed form for the forward) stay within
            // what the Infinity Cache serves -- beyond that the one-read sweep path wins.
            const int64_t stream_budget = (int64_t)env_int("GRAVHMC_RESIDENT_STREAM_MB", 320) << 20;
            if (env_int("GRAVHMC_RESIDENT_STREAM", 1) == 0 || 2 * c->ld * c->M * 8 > stream_budget) return false;
            if (c->wv.on && wavelet_dense_form(c) != GH_OK) return false;
            r.split = false;
            r.stream = true;
            r.ct = 0;
            int lc = cpw;
            while (lc > 0 && resident_lds_doubles(c->ld, cpw, 1, lc, false) * sizeof(double) > (size_t)lds_max) --lc;
            r.lds_cols = lc;
            lds = resident_lds_doubles(c->ld, cpw, 1, lc, false) * sizeof(double);
            if (lds > (size_t)lds_max) return false;
        }
    }
    r.lds = lds;
    resident_fn f = resident_for(r.rc, r.ct);
    if (!f) return false;
    if (allow_dynamic_lds(reinterpret_cast<const void *>(f), lds) != hipSuccess) {
        (void)hipGetLastError();
        return false;
    }
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(f), RES_THREADS,
                                                     lds) != hipSuccess || per_cu < 1 ||
        (int64_t)per_cu * c->cus < r.nwg) {
        (void)hipGetLastError();
        return false;
    }
    r.state = 1;
    return true;
}

// resident_plan + what depends on the regulariser set at the moment
static bool resident_usable(gh_ctx *c)
{
    if (!resident_plan(c)) return false;
    const bool stencil = c->reg_kind == 1 || c->reg_kind == 3;
    return !stencil || resident_stencil_fits(c->rs.cpw);
}

// One launch of the resident chain kernel: K trajectories of C chains (chain_of[k], nullptr: all
// chain 0) whose current models are the rows of x_dev.  GH_RESIDENT_ABORTED: the kernel gave up
// waiting for its workgroups, nothing was changed.
struct ResLaunch {
    int C = 1, K = 0;
    const int *chain_of = nullptr, *L = nullptr;
    const double *p0s = nullptr, *us = nullptr;
    double dt = 0.0;
    int64_t stop_at_accepts = 0, accept_count0 = 0;
    double *x_dev = nullptr, *gcur_dev = nullptr, *ucur_dev = nullptr;
    int have_state = 0;
    bool want_x = false;
};

static int resident_launch(gh_ctx *c, const ResLaunch &q, int *accepted, double *out5s, int h_run[4])
{
    gh_ctx::Resident &r = c->rs;
    const size_t M = (size_t)c->M;
    const int K = q.K;
    HIPCHK(c, hipSetDevice(c->device));
    const size_t lds = resident_lds_doubles(c->ld, r.cpw, q.C, r.lds_cols, r.split) * sizeof(double);
    if (q.C < 1 || q.C > RES_MAX_CHAINS || lds > (size_t)r.lds_max)
        return fail(c, GH_ERR_ARG, "resident chain kernel: %d chains do not fit the LDS", q.C);
    if (!resident_usable(c))
        return fail(c, GH_ERR_UNSUPPORTED, "resident chain kernel: this regulariser needs the sweep path at %d cells "
                                           "per workgroup", r.cpw);
    HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(resident_for(r.rc, r.ct)), lds));
    if (!r.slabg) {
        // (+8 rows / entries: the abort test announces one phantom workgroup per cluster)
        TRY(dalloc(c, &r.slabg, (size_t)(r.nwg + 8) * (size_t)c->ld * 2));
        TRY(dalloc(c, &r.xslabg, 2 * (size_t)RES_CLUSTERS * (size_t)c->ld * 2));
        TRY(dalloc(c, &r.dclg, (size_t)RES_CLUSTERS * (size_t)c->ld * 2));
        TRY(dalloc(c, &r.scalg, (size_t)(r.nwg + 8) * 8));
        TRY(dalloc(c, &r.xscalg, (size_t)RES_CLUSTERS * 8));
        TRY(dalloc(c, &r.xccg, (size_t)r.nwg + 8));
        TRY(dalloc(c, &r.xpub, 2 * M));
        TRY(dalloc(c, &r.abort_w, 4));
        TRY(dalloc(c, &r.n_run, 4));
        if (env_int("GRAVHMC_RESIDENT_TIMING", 0)) TRY(dalloc(c, &r.dbg, 32));
        if (!r.ev0) {
            HIPCHK(c, hipEventCreate(&r.ev0));
            HIPCHK(c, hipEventCreate(&r.ev1));
        }
    }
    if (K > r.Kcap || !r.L) {
        // grown rarely (the host batches a fixed number of trajectories per call); the old blocks
        // stay in the context's allocation list until gh_destroy
        const int cap = std::max(K, 32);
        r.L = r.accepted = r.chain = nullptr;
        r.p0s = r.us = r.out5s = r.xacc = nullptr;
        TRY(dalloc(c, &r.L, (size_t)cap));
        TRY(dalloc(c, &r.chain, (size_t)cap));
        TRY(dalloc(c, &r.accepted, (size_t)cap));
        TRY(dalloc(c, &r.p0s, (size_t)cap * M, false));
        TRY(dalloc(c, &r.us, (size_t)cap));
        TRY(dalloc(c, &r.out5s, (size_t)cap * 5));
        TRY(dalloc(c, &r.xacc, (size_t)cap * M, false));
        r.Kcap = cap;
    }
    int64_t steps = 0;
    for (int k = 0; k < K; ++k) steps += q.L[k];
    if (r.granules_dirty || (uint64_t)r.tag + (uint64_t)steps + (uint64_t)q.C + 2 > 0xf0000000ull ||
        (uint64_t)r.tagE + (uint64_t)K + (uint64_t)q.C + 2 > 0xf0000000ull) {
        // 32-bit tags about to wrap, or an aborted launch left granules carrying tags this launch
        // would use again: start the count again on zeroed granules
        HIPCHK(c, hipMemsetAsync(r.slabg, 0, (size_t)(r.nwg + 8) * (size_t)c->ld * 2 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.xslabg, 0, 2 * (size_t)RES_CLUSTERS * (size_t)c->ld * 2 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.dclg, 0, (size_t)RES_CLUSTERS * (size_t)c->ld * 2 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.scalg, 0, (size_t)(r.nwg + 8) * 8 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.xscalg, 0, (size_t)RES_CLUSTERS * 8 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.xccg, 0, ((size_t)r.nwg + 8) * sizeof(ghk::u64), c->stream));
        r.tag = r.tagE = 0;
        r.granules_dirty = false;
    }
    HIPCHK(c, hipMemsetAsync(r.abort_w, 0, 4 * sizeof(unsigned), c->stream));
    // (K = 0: only the potential and gradient at the chains' current samples are evaluated)
    if (K > 0) {
        HIPCHK(c, hipMemcpyAsync(r.p0s, q.p0s, (size_t)K * M * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(r.us, q.us, (size_t)K * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(r.L, q.L, (size_t)K * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    if (q.chain_of && K > 0)
        HIPCHK(c, hipMemcpyAsync(r.chain, q.chain_of, (size_t)K * sizeof(int), hipMemcpyHostToDevice, c->stream));
    ResArgs a{};
    a.G = c->G;
    a.Gl = c->wv.on ? c->wv.F : c->G;
    a.ld = c->ld;
    a.N = c->N;
    a.M = c->M;
    a.cols_per_wg = r.cpw;
    a.split = r.split ? 1 : 0;
    a.stream = r.stream ? 1 : 0;
    a.lds_cap = r.lds_cols;
    a.nwg = r.nwg;
    // test hook: the workgroups wait for partners that do not exist, time out and abort
    if (env_int("GRAVHMC_RESIDENT_TEST_ABORT", 0)) a.nwg += 8;
    a.try_local = env_int("GRAVHMC_RESIDENT_LOCAL", 1);
    a.gfix = c->have_fix ? c->gfix : nullptr;
    a.dobs_c = c->dobs_c;
    a.low = c->low;
    a.high = c->high;
    a.kind = c->reg_kind;
    a.nz = c->shape[0];
    a.ny = c->shape[1];
    a.nx = c->shape[2];
    a.alpha = c->alpha;
    a.beta = c->beta;
    a.mwapr = c->mwapr;
    a.wm2 = c->wm2;
    a.C = q.C;
    a.chain = q.chain_of ? r.chain : nullptr;
    a.x_cur = q.x_dev;
    a.gcur_io = q.gcur_dev;
    a.ucur_io = q.ucur_dev;
    a.have_state = q.have_state;
    a.K = K;
    a.L = r.L;
    a.p0s = r.p0s;
    a.us = r.us;
    a.dt = q.dt;
    a.stop_at_accepts = q.stop_at_accepts;
    a.accept_count0 = q.accept_count0;
    a.accepted = r.accepted;
    a.out5s = r.out5s;
    a.xacc = q.want_x ? r.xacc : nullptr;
    a.n_run = r.n_run;
    a.slabg = r.slabg;
    a.xslabg = r.xslabg;
    a.dclg = r.dclg;
    a.scalg = r.scalg;
    a.xscalg = r.xscalg;
    a.xccg = r.xccg;
    a.xpub = r.xpub;
    a.tag0 = r.tag;
    a.tagE0 = r.tagE;
    a.abort_w = r.abort_w;
    a.dbg = r.dbg;
    if (c->prof) HIPCHK(c, hipEventRecord(r.ev0, c->stream));
    // A plain launch: the grid was checked against the occupancy query in resident_plan (one
    // workgroup per CU by its LDS request), which is all hipLaunchCooperativeKernel would add;
    // residency itself is the same for both, and every wait inside the kernel is bounded.
    hipLaunchKernelGGL(resident_for(r.rc, r.ct), dim3(r.nwg), dim3(RES_THREADS), lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    if (c->prof) HIPCHK(c, hipEventRecord(r.ev1, c->stream));
    unsigned h_sync[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(h_sync, r.abort_w, sizeof h_sync, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_run, r.n_run, 4 * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (K > 0) {
        HIPCHK(c, hipMemcpyAsync(accepted, r.accepted, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(out5s, r.out5s, (size_t)K * 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_sync[0] != 0u) {
        // A workgroup waited 2 s for the others: they were not all resident (another process holding
        // compute units of this device).  Nothing of the chain state was written; the caller's batch
        // is run on the sweep-per-launch path.  A transient stall does not downgrade the context:
        // the next batch tries the resident kernel again (on cleared exchange buffers); after three
        // aborted launches the context stays on the sweep path.
        r.aborts += 1;
        r.granules_dirty = true;
        const bool for_good = r.aborts >= 3;
        if (for_good) r.state = -1;
        fprintf(stderr, "libgravhmc: resident chain kernel timed out waiting for its workgroups (%d of 3); "
                        "%s on the sweep-per-launch path\n", r.aborts,
                for_good ? "continuing for good" : "running this batch");
        return GH_RESIDENT_ABORTED;
    }
    r.tag += (unsigned)h_run[1];
    r.tagE += (unsigned)h_run[2];
    r.launches += 1;
    r.evals += h_run[1];
    if (c->prof) {
        float t = 0.f;
        HIPCHK(c, hipEventElapsedTime(&t, r.ev0, r.ev1));
        c->prof_ms_acc += t;
        c->prof_res_evals += h_run[1];
    }
    return GH_OK;
}

// K trajectories of the context's chain in one launch (same contract as gh_chain_run)
static int chain_run_resident(gh_ctx *c, int K, const int *L, const double *p0s, const double *us, double dt,
                              int64_t stop_at_accepts, int64_t record_from, int *accepted, double *out5s,
                              double *x_out, int *n_run)
{
    gh_ctx::Resident &r = c->rs;
    const size_t M = (size_t)c->M;
    ResLaunch q;
    q.K = K;
    q.L = L;
    q.p0s = p0s;
    q.us = us;
    q.dt = dt;
    q.stop_at_accepts = stop_at_accepts;
    q.accept_count0 = c->accept_count;
    q.x_dev = c->xb[c->xcur];
    q.want_x = x_out != nullptr || c->ring != nullptr;
    int h_run[4] = {0, 0, 0, 0};
    TRY(resident_launch(c, q, accepted, out5s, h_run));
    *n_run = h_run[0];
    for (int k = 0; k < h_run[0]; ++k) {
        if (!accepted[k]) continue;
        c->accept_count += 1;
        if (c->ring && c->accept_count > record_from) {
            ring_store_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(
                r.xacc + (size_t)k * M, c->weighted ? c->wm : nullptr, c->M,
                c->ring + (size_t)c->ring_next * M);
            c->ring_next = (c->ring_next + 1) % c->ring_K;
            c->ring_count += 1;
        }
        if (x_out)
            HIPCHK(c, hipMemcpyAsync(x_out + (size_t)k * M, r.xacc + (size_t)k * M, M * sizeof(double),
                                     hipMemcpyDeviceToHost, c->stream));
    }
    // the per-launch state (d, r, scalars of the current sample) is behind x now: whoever reads it
    // next brings it up to date (chain_state_fresh) -- not every batch, 150 us of launches and a
    // round trip each
    c->spec_valid = c->pn_valid = false;
    c->st_stale = true;
    return GH_OK;
}
