// libgravhmc host side: the chains of a batch in LOCK-STEP inside the resident batch kernel (resbatch.hip.h).
// Included once by gravhmc.hip.
#pragma once

typedef void (*resbatch_fn_t)(ResBatchArgs);

template <int KS>
static resbatch_fn_t resbatch_for_ks(int nt, bool greg)
{
    if (greg) return nt <= 1 ? resident_batch_kernel<KS, 1, true> : resident_batch_kernel<KS, 2, true>;
    return nt <= 1 ? resident_batch_kernel<KS, 1, false> : resident_batch_kernel<KS, 2, false>;
}

// ks: k-steps (4 rows) per wave of the adjoint, 32 ks >= ld; compiled for ld <= 640.  greg: the adjoint's
// operand in registers (compressed forward: LDS holds another matrix)
static resbatch_fn_t resbatch_for(int ks, int nt, bool greg)
{
    switch (ks) {
    case 5: return resbatch_for_ks<5>(nt, greg);
    case 10: return resbatch_for_ks<10>(nt, greg);
    case 15: return resbatch_for_ks<15>(nt, greg);
    case 20: return resbatch_for_ks<20>(nt, greg);
    }
    return nullptr;
}

// Can the batch of C chains run in lock-step?  The problem must fit the resident chain kernel with every
// column of a workgroup in LDS (no split / stream mode), at most 32 columns per workgroup and 640 rows.
static bool resbatch_plan(gh_ctx *c, int C)
{
    gh_ctx::Resident &r = c->rs;
    gh_ctx::Resident::LockStep &b = r.ls;
    b.on = false;
    if (env_int("GRAVHMC_RESIDENT_BATCH", 1) == 0 || C < 2 || C > 16) return false;
    if (!resident_usable(c) || r.split || r.stream || r.cpw > 32 || c->ld > 640 || c->ld % 16 != 0) return false;
    b.ks = (int)((c->ld + 159) / 160) * 5;
    b.nt = (r.cpw + 15) / 16;
    b.C = C;
    b.lds = resbatch_lds_doubles(c->ld, r.cpw, c->have_fix, c->wv.on) * sizeof(double);
    if (b.lds > (size_t)r.lds_max) return false;
    resbatch_fn_t f = resbatch_for(b.ks, b.nt, c->wv.on);
    if (!f) return false;
    int per_cu = 0;
    if (allow_dynamic_lds(reinterpret_cast<const void *>(f), b.lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(f), RB_THREADS, b.lds) != hipSuccess ||
        per_cu < 1 || (int64_t)per_cu * c->cus < r.nwg) {
        (void)hipGetLastError();
        return false;
    }
    const size_t M = (size_t)c->M, ldx = (size_t)c->ld + RB_XROWS;
    if (dalloc(c, &b.slabd, (size_t)(r.nwg + 8) * ldx * 16) != GH_OK || dalloc(c, &b.flagg, (size_t)r.nwg + 8) != GH_OK ||
        dalloc(c, &b.xslabg, 2 * (size_t)RES_CLUSTERS * ldx * 16) != GH_OK ||
        dalloc(c, &b.dclg, (size_t)RES_CLUSTERS * ldx * 16) != GH_OK || dalloc(c, &b.xccg, (size_t)r.nwg + 8) != GH_OK ||
        dalloc(c, &b.xpub, 2 * 16 * M) != GH_OK || dalloc(c, &b.xs, 16 * M) != GH_OK || dalloc(c, &b.ps, 16 * M) != GH_OK ||
        dalloc(c, &b.pst, 16 * M) != GH_OK || dalloc(c, &b.cst, 16 * RB_CST) != GH_OK || dalloc(c, &b.n_io, 128) != GH_OK ||
        dalloc(c, &r.abort_w, 4) != GH_OK)
        return false;
    if (env_int("GRAVHMC_RESIDENT_TIMING", 0) && dalloc(c, &b.dbg, 32) != GH_OK) return false;
    if (!r.ev0 && (hipEventCreate(&r.ev0) != hipSuccess || hipEventCreate(&r.ev1) != hipSuccess)) return false;
    // (a new batch: nothing in flight)
    if (hipMemsetAsync(b.cst, 0, 16 * RB_CST * sizeof(double), c->stream) != hipSuccess) return false;
    for (bool &f2 : b.active) f2 = false;
    b.on = true;
    return true;
}

// Potential and gradient at the chains' current samples, once per gh_batch_init: a launch of the resident
// chain kernel without trajectories (its start-up evaluates every chain and keeps the results on the device).
static int resbatch_state(gh_ctx *c)
{
    gh_ctx::Resident &r = c->rs;
    if (r.b_state) return GH_OK;
    ResLaunch q;
    q.C = r.ls.C;
    q.K = 0;
    q.x_dev = r.bx;
    q.gcur_dev = r.bg;
    q.ucur_dev = r.bu;
    q.have_state = 0;
    int h_run[4] = {0, 0, 0, 0};
    TRY(resident_launch(c, q, nullptr, nullptr, h_run));
    r.b_state = true;
    return GH_OK;
}

// One launch: up to T further trajectories of every chain (lists chain-major: element (ch, t) at ch * T + t).
// carry: end as soon as a chain has nothing left to start (trajectories in flight continue in the next
// call); otherwise every list is run to its end.  Results of chain ch in slots ch * Tout + i, Tout = T + 1
// with carry.  GH_RESIDENT_ABORTED: the kernel gave up waiting for its workgroups, nothing was changed.
static int resbatch_launch(gh_ctx *c, int T, const int *L, const double *const *p0rows, const double *p0flat,
                           const double *us, double dt, bool carry, int *accepted, double *out5s, double *x_out,
                           int *n_started, int *n_done)
{
    gh_ctx::Resident &r = c->rs;
    gh_ctx::Resident::LockStep &b = r.ls;
    const size_t M = (size_t)c->M;
    const int C = b.C, Tout = carry ? T + 1 : T;
    const size_t ldx = (size_t)c->ld + RB_XROWS;
    HIPCHK(c, hipSetDevice(c->device));
    const int K = C * T, Kout = C * std::max(Tout, 1);
    if (std::max(K, Kout) > b.cap) {
        const int cap = std::max(std::max(K, Kout), 32);
        b.L = b.accepted = nullptr;
        b.p0s = b.us = b.out5s = b.xacc = nullptr;
        TRY(dalloc(c, &b.L, (size_t)cap));
        TRY(dalloc(c, &b.accepted, (size_t)cap));
        TRY(dalloc(c, &b.p0s, (size_t)cap * M, false));
        TRY(dalloc(c, &b.us, (size_t)cap));
        TRY(dalloc(c, &b.out5s, (size_t)cap * 5));
        TRY(dalloc(c, &b.xacc, (size_t)cap * M, false));
        b.cap = cap;
    }
    int64_t steps = 0;
    for (int k = 0; k < K; ++k) steps += L[k] + 1;
    steps += 64 * 17;  // (what is in flight; generous)
    if (b.dirty || (uint64_t)b.tag + (uint64_t)steps + 2 > 0xf0000000ull || b.ltag > 0xf0000000u) {
        HIPCHK(c, hipMemsetAsync(b.flagg, 0, ((size_t)r.nwg + 8) * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(b.xslabg, 0, 2 * (size_t)RES_CLUSTERS * ldx * 16 * sizeof(ghk::u32x4), c->stream));
        HIPCHK(c, hipMemsetAsync(b.dclg, 0, (size_t)RES_CLUSTERS * ldx * 16 * sizeof(ghk::u32x4), c->stream));
        HIPCHK(c, hipMemsetAsync(b.xccg, 0, ((size_t)r.nwg + 8) * sizeof(ghk::u64), c->stream));
        b.tag = b.ltag = 0;
        b.dirty = false;
    }
    HIPCHK(c, hipMemsetAsync(r.abort_w, 0, 4 * sizeof(unsigned), c->stream));
    // Momenta: device rows chain-major (ch * T + t), like the lists.  Rows inside a block of gh_pinned_alloc go
    // straight from where they lie, adjacent ones in one copy (a sampler drawing into a ring of such rows: one
    // or two copies per chain); the others are gathered into ONE pinned staging buffer first (a few host threads:
    // 6 MB at C1 with 16 chains x 8 trajectories) -- 128 separate copies of pageable rows cost more than the
    // kernel they feed.
    if ((size_t)K * M > b.h_stage_n) {
        if (b.h_stage) HIPCHK(c, hipHostFree(b.h_stage));
        b.h_stage = nullptr;
        b.h_stage_n = (size_t)std::max(K, 32) * M;
        HIPCHK(c, hipHostMalloc((void **)&b.h_stage, b.h_stage_n * sizeof(double)));
    }
    if (K > 0) {
        std::vector<const double *> src((size_t)K);
        std::vector<char> direct((size_t)K, 0);
        int n_staged = 0;
        for (int k = 0; k < K; ++k) {
            src[(size_t)k] = p0flat ? p0flat + (size_t)k * M : p0rows[(size_t)k];
            const char *lo = (const char *)src[(size_t)k], *hi = lo + M * sizeof(double);
            for (const gh_ctx::Pinned &pm : c->pinned)
                if (lo >= pm.base && hi <= pm.base + pm.bytes) {
                    direct[(size_t)k] = 1;
                    break;
                }
            n_staged += direct[(size_t)k] ? 0 : 1;
        }
        b.rows_direct += K - n_staged;
        b.rows_staged += n_staged;
        if (n_staged > 0) {
            auto stage_rows = [&](int k0, int k1) {
                for (int k = k0; k < k1; ++k)
                    if (!direct[(size_t)k]) memcpy(b.h_stage + (size_t)k * M, src[(size_t)k], M * sizeof(double));
            };
            const int nthr = (size_t)n_staged * M * sizeof(double) >= ((size_t)1 << 20) ? std::min(4, K) : 1;
            if (nthr > 1) {
                std::vector<std::thread> pool;
                for (int i = 1; i < nthr; ++i) pool.emplace_back(stage_rows, (int)((int64_t)K * i / nthr), (int)((int64_t)K * (i + 1) / nthr));
                stage_rows(0, K / nthr);
                for (std::thread &th : pool) th.join();
            } else {
                stage_rows(0, K);
            }
        }
        // one copy per run of rows that are adjacent at their source
        for (int k0 = 0; k0 < K;) {
            int k1 = k0 + 1;
            const double *from = direct[(size_t)k0] ? src[(size_t)k0] : b.h_stage + (size_t)k0 * M;
            while (k1 < K && direct[(size_t)k1] == direct[(size_t)k0] &&
                   (direct[(size_t)k0] ? src[(size_t)k1] == src[(size_t)k1 - 1] + M : true))
                ++k1;
            HIPCHK(c, hipMemcpyAsync(b.p0s + (size_t)k0 * M, from, (size_t)(k1 - k0) * M * sizeof(double), hipMemcpyHostToDevice,
                                     c->stream));
            k0 = k1;
        }
        HIPCHK(c, hipMemcpyAsync(b.us, us, (size_t)K * sizeof(double), hipMemcpyHostToDevice, c->stream));
        HIPCHK(c, hipMemcpyAsync(b.L, L, (size_t)K * sizeof(int), hipMemcpyHostToDevice, c->stream));
    }
    ResBatchArgs a{};
    a.G = c->G;
    a.Gl = c->wv.on ? c->wv.F : c->G;
    a.ld = c->ld;
    a.N = c->N;
    a.M = c->M;
    a.cols_per_wg = r.cpw;
    a.nwg = r.nwg;
    if (env_int("GRAVHMC_RESBATCH_TEST_ABORT", 0)) a.nwg += 8;  // test hook: partners that do not exist
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
    a.C = C;
    a.T = T;
    a.L = b.L;
    a.p0s = b.p0s;
    a.us = b.us;
    a.dt = dt;
    a.stop_any = carry ? 1 : 0;
    a.x_cur = r.bx;
    a.g_cur = r.bg;
    a.u_cur = r.bu;
    a.xs_io = b.xs;
    a.ps_io = b.ps;
    a.pst_io = b.pst;
    a.cst_io = b.cst;
    a.Tout = std::max(Tout, 1);
    a.accepted = b.accepted;
    a.out5s = b.out5s;
    a.xacc = x_out ? b.xacc : nullptr;
    a.n_io = b.n_io;
    a.slabd = b.slabd;
    a.flagg = b.flagg;
    a.xslabg = b.xslabg;
    a.dclg = b.dclg;
    a.xccg = b.xccg;
    a.xpub = b.xpub;
    a.tag0 = b.tag;
    a.ltag = b.ltag + 1u;
    a.abort_w = r.abort_w;
    a.dbg = b.dbg;
    resbatch_fn_t f = resbatch_for(b.ks, b.nt, c->wv.on);
    // (gh_set_data may have added or removed the fixed part of the data term since the batch was planned)
    b.lds = resbatch_lds_doubles(c->ld, r.cpw, c->have_fix, c->wv.on) * sizeof(double);
    if (b.lds > (size_t)r.lds_max) return fail(c, GH_ERR_UNSUPPORTED, "resident batch kernel: the problem no longer fits the LDS");
    HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(f), b.lds));
    if (c->prof) HIPCHK(c, hipEventRecord(r.ev0, c->stream));
    // (a plain launch: the grid was checked against the occupancy query in resbatch_plan; every wait inside is bounded)
    hipLaunchKernelGGL(f, dim3(r.nwg), dim3(RB_THREADS), b.lds, c->stream, a);
    HIPCHK(c, hipGetLastError());
    if (c->prof) HIPCHK(c, hipEventRecord(r.ev1, c->stream));
    unsigned h_sync[4] = {0, 0, 0, 0};
    int h_n[128];
    HIPCHK(c, hipMemcpyAsync(h_sync, r.abort_w, sizeof h_sync, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h_n, b.n_io, sizeof h_n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    b.ltag += 1u;
    if (h_sync[0] != 0u) {
        b.aborts += 1;
        b.dirty = true;
        fprintf(stderr, "libgravhmc: resident batch kernel timed out waiting for its workgroups; the chains take turns in "
                        "the resident chain kernel from here on\n");
        return GH_RESIDENT_ABORTED;
    }
    const int lock_steps = h_n[32];
    b.tag += (unsigned)lock_steps;
    b.launches += 1;
    b.lock_steps += lock_steps;
    // results
    std::vector<int> acc((size_t)Kout);
    std::vector<double> o5((size_t)Kout * 5);
    HIPCHK(c, hipMemcpyAsync(acc.data(), b.accepted, (size_t)Kout * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(o5.data(), b.out5s, (size_t)Kout * 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // accepted models: one copy per chain (its done slots are adjacent) into pinned staging, rows picked from there --
    // a copy per accepted row into the caller's pageable array costs ~15 us each, hundreds per call
    bool want_rows = false;
    for (int ch = 0; ch < C; ++ch) {
        const int nd = h_n[16 + ch];
        if (nd > a.Tout || h_n[ch] > T)
            return fail(c, GH_ERR_HIP, "resident batch kernel: chain %d reports %d results in %d slots (%d of %d started)", ch, nd,
                        a.Tout, h_n[ch], T);
        if (!x_out) continue;
        int last = -1;
        for (int i = 0; i < nd; ++i)
            if (acc[(size_t)ch * a.Tout + i]) last = i;
        if (last < 0) continue;
        if (!want_rows && (size_t)Kout * M > b.h_xstage_n) {
            if (b.h_xstage) HIPCHK(c, hipHostFree(b.h_xstage));
            b.h_xstage = nullptr;
            b.h_xstage_n = (size_t)std::max(Kout, 32) * M;
            HIPCHK(c, hipHostMalloc((void **)&b.h_xstage, b.h_xstage_n * sizeof(double)));
        }
        want_rows = true;
        const size_t s0 = (size_t)ch * a.Tout;
        HIPCHK(c, hipMemcpyAsync(b.h_xstage + s0 * M, b.xacc + s0 * M, (size_t)(last + 1) * M * sizeof(double), hipMemcpyDeviceToHost,
                                 c->stream));
    }
    if (want_rows) HIPCHK(c, hipStreamSynchronize(c->stream));
    for (int ch = 0; ch < C; ++ch) {
        const int nd = h_n[16 + ch];
        for (int i = 0; i < nd; ++i) {
            const size_t slot = (size_t)ch * a.Tout + i;
            accepted[slot] = acc[slot];
            memcpy(out5s + slot * 5, o5.data() + slot * 5, 5 * sizeof(double));
            if (x_out && acc[slot]) memcpy(x_out + slot * M, b.h_xstage + slot * M, M * sizeof(double));
        }
        if (n_started) n_started[ch] = h_n[ch];
        if (n_done) n_done[ch] = nd;
        b.active[ch] = h_n[64 + ch] != 0;
        b.lost += h_n[48 + ch];
        b.chain_steps += h_n[80 + ch];
    }
    if (c->prof) {
        float t = 0.f;
        HIPCHK(c, hipEventElapsedTime(&t, r.ev0, r.ev1));
        c->prof_ms_acc += t;
        c->prof_res_evals += lock_steps;
    }
    return GH_OK;
}

// The kernel gave up: the trajectories in flight start again from their chains' current samples (their own
// momentum, length and variate were kept on the device) in front of the new lists, on the chains-take-turns
// kernel.  Fills the host copies the caller needs to build that launch.
static int resbatch_inflight(gh_ctx *c, std::vector<int> &chains, std::vector<int> &L, std::vector<double> &us,
                             std::vector<double> &p0s)
{
    gh_ctx::Resident::LockStep &b = c->rs.ls;
    const size_t M = (size_t)c->M;
    std::vector<double> cst(16 * RB_CST);
    TRY(d2h(c, cst.data(), b.cst, cst.size()));
    for (int ch = 0; ch < b.C; ++ch)
        if (cst[(size_t)RB_CST * ch] != 0.0) {
            chains.push_back(ch);
            L.push_back((int)cst[(size_t)RB_CST * ch + 2]);
            us.push_back(cst[(size_t)RB_CST * ch + 3]);
            p0s.resize(p0s.size() + M);
            TRY(d2h(c, p0s.data() + p0s.size() - M, b.pst + (size_t)ch * M, M));
        }
    HIPCHK(c, hipMemsetAsync(b.cst, 0, 16 * RB_CST * sizeof(double), c->stream));
    for (bool &f : b.active) f = false;
    return GH_OK;
}
