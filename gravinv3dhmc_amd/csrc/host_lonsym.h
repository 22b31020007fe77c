// libgravhmc host side: detection and construction of the shift-invariant store for regular
// spherical grids (lonsym.hip.h).  Included once by gravhmc.hip.
#pragma once

struct LonSymHost {
    bool on = false;
    int n = 0, na = 0, nc = 0, SW = 0, AG = 0, KB = 0;
    int64_t ldT = 0;
    double *T = nullptr;
    int *slot_first = nullptr, *slot_x = nullptr, *xslot = nullptr, *xptr = nullptr, *xobs = nullptr, *lds_of = nullptr, *a_of = nullptr,
        *m_of = nullptr;
    int n_xslots = 0, max_extra = 0;  // max_extra: most observations a slot holds beyond its first
    long long *dbg = nullptr;  // GRAVHMC_LONSYM_TIMING: per-phase clocks of one workgroup
    // the same store in the longitude-harmonic domain (lonsymh.hip.h): default where it applies
    bool harm = false;
    int nf = 0, hgrid = 0, rw = 1;   // rw: cell rows a workgroup of the pass works on at once
    ghk::d2 *That = nullptr, *tw = nullptr, *Rhat = nullptr, *Dpart = nullptr;
    // the harmonic store as streaming passes over T^ (lonsymw.hip.h): grids beyond the register form's limits
    bool wide = false, direct_ok = false;
    int wfwd = 0;                          // variant of the forward product's kernel
    // north-south mirror of the streamed form: one row of T^ per pair of mirrored cell rows (lonsymw.hip.h)
    bool wmirror = false;
    int witems = 0;
    int *item_c = nullptr, *item_c2 = nullptr, *amir = nullptr;
    int nfp = 0, wbreak = 0;               // pitch of a row of T^ / R^ / D^ in the streamed form (complex entries)
    int wgrid = 0, wparts = 0, wrows = 0;  // workgroups of the sweep; parts of the forward product, cell rows per part
    ghk::d2 *Xhat = nullptr;
    size_t wlds = 0;
    size_t hlds = 0;
    // the epilogue in one launch behind the sweep (lonsymh_epilogue_kernel): default with the harmonic form
    bool fused = false;
    bool post_pending = false;       // the sweep's D^ partials have not been turned into d yet
    const double *rhat_of = nullptr; // the residual vector R^ was last computed from (by the fused epilogue)
    double *post_slab = nullptr, *post_dsum = nullptr;
    unsigned long long *csum = nullptr;
    unsigned *epi_abort = nullptr;
    unsigned epi_tag = 0;
    // the harmonic pass as one persistent launch per batch of trajectories (lonres.hip.h, host_lonres.h)
    struct Res {
        int state = 0;  // 0 not planned yet, 1 usable, -1 not applicable
        size_t lds = 0;
        ghk::d2 *slab = nullptr, *mhat = nullptr, *rhatg = nullptr;
        ghk::u64 *flagg = nullptr, *xccg = nullptr;
        ghk::u32x4 *xslabg = nullptr, *clsg = nullptr, *scalg = nullptr, *ppg = nullptr;
        unsigned *abort_w = nullptr;
        unsigned tag = 0, tagE = 0, ltag = 0;
        bool dirty = false;
        int Kcap = 0;
        int *L = nullptr, *accepted = nullptr, *n_run = nullptr;
        double *p0s = nullptr, *us = nullptr, *out5s = nullptr, *xacc = nullptr, *ucur = nullptr, *xpub = nullptr;
        double *h_stage = nullptr;
        size_t h_stage_n = 0;
        int64_t launches = 0, evals = 0, trajectories = 0;
        int aborts = 0;
        long long *dbg = nullptr;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
    } res;
    size_t lds = 0;
    int grid = 0, items = 1, W = 8, thr = 1024;  // items: work items per wave, W: longitudes per work item (instantiation of the kernel)
    std::string why;  // why the geometry does not qualify (gh_last_error text)
};

typedef void (*lonsym_fn_t)(LonSymGeom, SweepArgs, const double *);
static lonsym_fn_t lonsym_fn(int items, int W, int T)
{
    if (W == 16) return T == 512 ? lonsym_sweep_kernel<1, 16, 512> : lonsym_sweep_kernel<1, 16, 1024>;
    return items <= 1 ? lonsym_sweep_kernel<1, 8, 1024> : items <= 2 ? lonsym_sweep_kernel<2, 8, 1024> : lonsym_sweep_kernel<4, 8, 1024>;
}

typedef void (*lonsymw_fn_t)(LonWideGeom, SweepArgs, const double *);
static lonsymw_fn_t lonsymw_sweep_fn(int nf)
{
    return nf <= LW_THREADS ? lonsymw_sweep_kernel<1> : nf <= 2 * LW_THREADS ? lonsymw_sweep_kernel<2> : lonsymw_sweep_kernel<3>;
}

typedef void (*lonsymh_fn_t)(LonHarmGeom, SweepArgs, const double *);
static lonsymh_fn_t lonsymh_fn(int rw)
{
    return rw <= 1 ? lonsymh_sweep_kernel<1> : rw == 2 ? lonsymh_sweep_kernel<2> : rw == 3 ? lonsymh_sweep_kernel<3> : lonsymh_sweep_kernel<4>;
}

static LonSymGeom lonsym_geom(const gh_ctx *c)
{
    const LonSymHost &h = *c->ls;
    LonSymGeom g;
    g.n = h.n;
    g.na = h.na;
    g.nc = h.nc;
    g.SW = h.SW;
    g.AG = h.AG;
    g.KB = h.KB;
    g.ldT = h.ldT;
    g.T = h.T;
    g.slot_first = h.slot_first;
    g.n_xslots = h.n_xslots;
    g.xslot = h.xslot;
    g.xptr = h.xptr;
    g.xobs = h.xobs;
    g.lds_of = h.lds_of;
    g.N = c->N;
    g.dbg = h.dbg;
    return g;
}

// Does the geometry have the structure (header of lonsym.hip.h)?  Fills the host description and
// builds the table with the reference's adaptive engine; GH_ERR_UNSUPPORTED with the reason otherwise.
static int lonsym_build(gh_ctx *c)
{
    LonSymHost &h = *c->ls;
    h.on = false;
    auto no = [&](const char *why) {
        h.why = why;
        return fail(c, GH_ERR_UNSUPPORTED, "shift-invariant store: %s", why);
    };
    if (c->cell_kind != GH_CELL_TESSEROID) return no("tesseroid cells only");
    const int64_t M = c->M, N = c->N;
    std::vector<double> b((size_t)M * 6), lon((size_t)N), lat((size_t)N), hh((size_t)N);
    HIPCHK(c, hipMemcpyAsync(b.data(), c->bounds, sizeof(double) * b.size(), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(lon.data(), c->obs[0], sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(lat.data(), c->obs[1], sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(hh.data(), c->obs[2], sizeof(double) * (size_t)N, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    auto same_row = [&](int64_t j0, int64_t j1) {
        return memcmp(&b[(size_t)j0 * 6 + 2], &b[(size_t)j1 * 6 + 2], 4 * sizeof(double)) == 0;
    };
    // cells: rows of n cells that differ in their longitudes only, w = w0 + k dlon, n dlon = 360
    int64_t n = 1;
    while (n < M && same_row(0, n)) ++n;
    if (M % n != 0) return no("the cells do not form rows of equal length along the longitude");
    const double w0 = b[0], dlon = b[1] - b[0];
    if (!(dlon > 0) || std::fabs((double)n * dlon - 360.0) > 1e-9 * 360.0)
        return no("a row of cells does not cover the full circle of longitudes");
    const double tol = 1e-9 * dlon;
    for (int64_t j = 0; j < M; ++j) {
        const int64_t k = j % n;
        if (!same_row(j - k, j) || std::fabs(b[(size_t)j * 6] - (w0 + (double)k * dlon)) > tol ||
            std::fabs(b[(size_t)j * 6 + 1] - (w0 + (double)(k + 1) * dlon)) > tol)
            return no("the cell rows are not regular in longitude (same origin, same spacing)");
    }
    // observations: longitudes on the cells' spacing, classes of equal (latitude, height)
    std::vector<int> a_of((size_t)N), m_of((size_t)N);
    std::map<std::pair<uint64_t, uint64_t>, int> cls;
    std::vector<double> cl_lat, cl_h;
    const double lon_ref = lon[0];
    for (int64_t i = 0; i < N; ++i) {
        const long long m = std::llround((lon[(size_t)i] - lon_ref) / dlon);
        if (std::fabs(lon[(size_t)i] - lon_ref - (double)m * dlon) > tol)
            return no("the observation longitudes are not on the cells' longitude spacing");
        m_of[(size_t)i] = (int)(((m % n) + n) % n);
        uint64_t kb[2];
        memcpy(&kb[0], &lat[(size_t)i], 8);
        memcpy(&kb[1], &hh[(size_t)i], 8);
        auto it = cls.find({kb[0], kb[1]});
        if (it == cls.end()) {
            it = cls.emplace(std::make_pair(kb[0], kb[1]), (int)cl_lat.size()).first;
            cl_lat.push_back(lat[(size_t)i]);
            cl_h.push_back(hh[(size_t)i]);
        }
        a_of[(size_t)i] = it->second;
    }
    const int64_t na = (int64_t)cl_lat.size(), nc = M / n;
    h.n = (int)n;
    h.na = (int)na;
    h.nc = (int)nc;
    h.SW = (int)((n + 15) / 16 * 16 + 1);
    h.AG = (int)((na + 63) / 64);
    // 16 longitudes per work item where that still gives every SIMD a wave (and the items fit two per wave)
    h.W = (env_int("GRAVHMC_LONSYM_W", 16) == 16 && h.AG * ((n + 15) / 16) >= 4 && h.AG * ((n + 15) / 16) <= LS_WAVES) ? 16 : 8;
    h.KB = (int)((n + h.W - 1) / h.W);
    // (eight waves cover the items: half the threads, twice the registers -- the 16 + 16 + 16 doubles of
    // a work item's accumulators and table window then stay out of scratch)
    h.thr = (h.W == 16 && h.AG * h.KB <= 8 && n <= 512) ? 512 : 1024;
    const int steps = h.KB * h.W;
    h.SW = (int)((steps + 15) / 16 * 16 + 1);
    h.lds = sizeof(double) * ((size_t)(2 * na + 1) * h.SW + steps + 8 + (size_t)h.AG * h.KB * h.W + 32);
    // (the direct correlations of lonsym.hip.h hold a cell row's table and the residual grid in LDS; grids beyond that
    // run on the streamed harmonic form, lonsymw.hip.h, which only needs the transforms' tables there)
    const char *direct_why = nullptr;
    if (n > LS_THREADS) direct_why = "more than 1024 longitudes per cell row";
    else if (h.AG * h.KB > (h.thr / 64) * (h.W == 16 ? 1 : LS_MAXITEMS)) direct_why = "too many (class, longitude block) work items for one workgroup";
    else if (h.lds > 160 * 1024 - 512) direct_why = "a cell row's table and the residual grid do not fit the LDS";
    else if (na * n > (int64_t)8 * LS_THREADS) direct_why = "a cell row's table has more than 8192 entries";
    h.direct_ok = direct_why == nullptr;
    const bool wide_can = n <= LW_NMAX && n >= 2 && env_int("GRAVHMC_LONSYM_WIDE", 1) != 0;
    if (!h.direct_ok && !wide_can) return no(direct_why);
    if ((int64_t)na * n > 0x7fffffffLL / 4) return no("more than 2^29 (class, longitude) slots");
    // the table: every class at every shift against the cells of longitude index 0 (reference engine)
    const int64_t Np = na * n;
    h.ldT = (Np + 15) / 16 * 16;
    std::vector<double> so((size_t)Np * 3), sb((size_t)nc * 6);
    for (int64_t a = 0; a < na; ++a)
        for (int64_t d = 0; d < n; ++d) {
            so[(size_t)(a * n + d)] = lon_ref + (double)d * dlon;
            so[(size_t)(Np + a * n + d)] = cl_lat[(size_t)a];
            so[(size_t)(2 * Np + a * n + d)] = cl_h[(size_t)a];
        }
    for (int64_t cc = 0; cc < nc; ++cc) memcpy(&sb[(size_t)cc * 6], &b[(size_t)(cc * n) * 6], 6 * sizeof(double));
    double *d_so = nullptr, *d_sb = nullptr, *conv = nullptr;
    int *err_cell = nullptr;
    TessStats *stats = nullptr;
    TRY(dalloc(c, &h.T, (size_t)h.ldT * (size_t)nc, false));
    // (five temporaries of the build: released on every way out of this block)
    struct Tmp {
        void *p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
        ~Tmp() { for (void *q : p) if (q) (void)hipFree(q); }
    } tmp;
    HIPCHK(c, hipMalloc(&tmp.p[0], sizeof(double) * so.size()));
    HIPCHK(c, hipMalloc(&tmp.p[1], sizeof(double) * sb.size()));
    HIPCHK(c, hipMalloc(&tmp.p[2], sizeof(double) * 4 * (size_t)Np));
    HIPCHK(c, hipMalloc(&tmp.p[3], sizeof(int) * (size_t)nc));
    HIPCHK(c, hipMalloc(&tmp.p[4], sizeof(TessStats)));
    d_so = static_cast<double *>(tmp.p[0]);
    d_sb = static_cast<double *>(tmp.p[1]);
    conv = static_cast<double *>(tmp.p[2]);
    err_cell = static_cast<int *>(tmp.p[3]);
    stats = static_cast<TessStats *>(tmp.p[4]);
    HIPCHK(c, hipMemcpyAsync(d_so, so.data(), sizeof(double) * so.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_sb, sb.data(), sizeof(double) * sb.size(), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(err_cell, 0, sizeof(int) * (size_t)nc, c->stream));
    HIPCHK(c, hipMemsetAsync(stats, 0, sizeof(TessStats), c->stream));
    tess_convert_kernel<<<dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, c->stream>>>(
        d_so, d_so + Np, d_so + 2 * Np, Np, conv, conv + Np, conv + 2 * Np, conv + 3 * Np);
    const int64_t total = h.ldT * nc;
    tess_gz_kernel<<<dim3((unsigned)std::min<int64_t>((total + 63) / 64, 1 << 24)), dim3(64), 0, c->stream>>>(
        conv, conv + Np, conv + 2 * Np, conv + 3 * Np, d_sb, Np, nc, h.ldT, c->ratio, h.T, err_cell, stats);
    HIPCHK(c, hipGetLastError());
    TessStats hs;
    std::vector<int> herr((size_t)nc);
    HIPCHK(c, hipMemcpyAsync(&hs, stats, sizeof hs, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(herr.data(), err_cell, sizeof(int) * (size_t)nc, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (hs.overflow) return fail(c, GH_ERR_OVERFLOW, "tesseroid stack overflow (> %d entries)", TESS_STACK);
    // a cell of longitude index 0 that could not be divided further stands for its whole row of n cells:
    // the count gh_kernel_stats reports is the dense build's (the reference's warning, potential.py:134)
    for (int v : herr)
        if (v != 0) c->warn_cells += n;
    c->leaves = (int64_t)hs.leaves;
    // slots (a, m) -> observations, ascending: the first one per slot, and the further ones of the few
    // slots that hold several (duplicated longitudes); LDS offset of every observation's slot
    std::vector<int> sfirst((size_t)Np, -1), xslot, xptr(1, 0), xobs, ldsof((size_t)N);
    {
        std::vector<std::vector<int>> more((size_t)Np);
        for (int64_t i = 0; i < N; ++i) {
            const size_t sl = (size_t)(a_of[(size_t)i] * n + m_of[(size_t)i]);
            if (sfirst[sl] < 0)
                sfirst[sl] = (int)i;
            else
                more[sl].push_back((int)i);
        }
        for (int64_t sl = 0; sl < Np; ++sl)
            if (!more[(size_t)sl].empty()) {
                xslot.push_back((int)sl);
                h.max_extra = std::max(h.max_extra, (int)more[(size_t)sl].size());
                xobs.insert(xobs.end(), more[(size_t)sl].begin(), more[(size_t)sl].end());
                xptr.push_back((int)xobs.size());
            }
    }
    h.n_xslots = (int)xslot.size();
    std::vector<int> slotx((size_t)Np, -1);
    for (size_t x = 0; x < xslot.size(); ++x) slotx[(size_t)xslot[x]] = (int)x;
    if (xslot.empty()) xslot.push_back(0);
    if (xobs.empty()) xobs.push_back(0);
    for (int64_t i = 0; i < N; ++i) ldsof[(size_t)i] = a_of[(size_t)i] * h.SW + m_of[(size_t)i];
    auto up = [&](int **dst, const std::vector<int> &src) -> int {
        TRY(dalloc(c, dst, src.size(), false));
        HIPCHK(c, hipMemcpyAsync(*dst, src.data(), sizeof(int) * src.size(), hipMemcpyHostToDevice, c->stream));
        return GH_OK;
    };
    TRY(up(&h.slot_first, sfirst));
    TRY(up(&h.slot_x, slotx));
    TRY(up(&h.xslot, xslot));
    TRY(up(&h.xptr, xptr));
    TRY(up(&h.xobs, xobs));
    TRY(up(&h.lds_of, ldsof));
    TRY(up(&h.a_of, a_of));
    TRY(up(&h.m_of, m_of));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    h.items = (h.AG * h.KB + h.thr / 64 - 1) / (h.thr / 64);
    if (h.direct_ok) HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(lonsym_fn(h.items, h.W, h.thr)), h.lds));
    h.grid = (int)std::min<int64_t>(nc, c->cus);
    if (env_int("GRAVHMC_LONSYM_TIMING", 0)) TRY(dalloc(c, &h.dbg, 8));
    // The harmonic form (lonsymh.hip.h): n / 2 + 1 frequencies in the lanes of a wave, the classes in four
    // groups of at most LH_AK, R^ and the transforms' scratch in LDS.  GRAVHMC_LONSYM_HARMONIC=0: the direct
    // correlations of lonsym.hip.h (also the fallback for geometries beyond these limits).
    h.harm = false;
    h.nf = (int)n / 2 + 1;
    // (one workgroup per CU at most -- its T^ rows and accumulators take the CU's registers -- every one with the
    // same number of cell rows, up to four of them at once)
    const int rp = (int)((nc + (int64_t)c->cus - 1) / (int64_t)c->cus);
    h.rw = std::min(rp, 4);
    h.hlds = lonsymh_lds_doubles((int)n, h.nf, (int)na, h.rw) * sizeof(double);
    const int force_wide = env_int("GRAVHMC_LONSYM_WIDE", 1) == 2;  // (2: the streamed form also where the register form applies)
    if (!force_wide && env_int("GRAVHMC_LONSYM_HARMONIC", 1) != 0 && h.nf <= 64 && na <= 4 * LH_AK && n <= 1024 && h.hlds <= 160 * 1024 - 512 &&
        allow_dynamic_lds(reinterpret_cast<const void *>(lonsymh_fn(h.rw)), h.hlds) == hipSuccess) {
        h.hgrid = (int)((nc + rp - 1) / rp);
        TRY(dalloc(c, &h.tw, (size_t)n, false));
        TRY(dalloc(c, &h.That, (size_t)nc * (size_t)na * (size_t)h.nf, false));
        TRY(dalloc(c, &h.Rhat, (size_t)na * (size_t)h.nf));
        TRY(dalloc(c, &h.Dpart, (size_t)h.hgrid * (size_t)na * (size_t)h.nf));
        lonsymh_twiddle_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>((int)n, h.tw);
        lonsymh_table_kernel<<<dim3((unsigned)(nc * na)), dim3(64), 0, c->stream>>>(h.T, h.ldT, (int)n, h.nf, (int)na, h.tw, h.That, h.nf);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        h.harm = true;
        // (the epilogue in one launch: built and tested, NOT the default -- measured at C4 it takes 21.9 us against
        // 12.7 + 4.1 + 7.4 for the three launches it replaces, 16.4 k steps/s against 17.3 k: the sum of the sweep's
        // 200 partials per class is latency-bound on 256 threads, and larger blocks could not all be resident
        // when eight chains run their epilogues at once)
        h.fused = env_int("GRAVHMC_LONSYM_FUSED", 0) != 0;
        if (h.fused) {
            TRY(dalloc(c, &h.csum, 2 * (size_t)na));
            TRY(dalloc(c, &h.epi_abort, 4));
        }
    } else {
        (void)hipGetLastError();
    }
    // The streamed harmonic form (lonsymw.hip.h) where the register form does not apply: no limit on the classes,
    // n <= 1024.  GRAVHMC_LONSYM_WIDE=0: off (the direct correlations, or a refusal), 2: also where the register form applies.
    h.wide = false;
    if (!h.harm && wide_can && (env_int("GRAVHMC_LONSYM_HARMONIC", 1) != 0 || force_wide || !h.direct_ok)) {
        // (rows of T^ start on 128-byte lines: a wave's 1 KB request then touches 8 of them, not 9)
        h.nfp = (h.nf + 7) / 8 * 8;
        // North-south mirror: a grid symmetric about the equator (cell rows and observation classes come in mirrored pairs)
        // has K(row c, class a) = K(mirror row, mirror class): ONE row of T^ per pair, every entry read serves both rows
        // -- half the table, half the bytes per step.  GRAVHMC_LW_MIRROR=0: off.
        std::vector<int> itc, itc2, amirv((size_t)na, -1);
        h.wmirror = false;
        if (env_int("GRAVHMC_LW_MIRROR", 1) != 0) {
            auto near = [](double x, double y) { return std::fabs(x - y) <= 1e-9 * std::max(1.0, std::max(std::fabs(x), std::fabs(y))); };
            bool ok = true;
            for (int64_t a = 0; a < na && ok; ++a) {
                for (int64_t b2 = 0; b2 < na; ++b2)
                    if (near(cl_lat[(size_t)b2], -cl_lat[(size_t)a]) && near(cl_h[(size_t)b2], cl_h[(size_t)a])) {
                        amirv[(size_t)a] = (int)b2;
                        break;
                    }
                ok = amirv[(size_t)a] >= 0;
            }
            for (int64_t a = 0; a < na && ok; ++a) ok = amirv[(size_t)amirv[(size_t)a]] == (int)a;
            std::vector<int> rmir((size_t)nc, -1);
            for (int64_t cc = 0; cc < nc && ok; ++cc) {
                const double *q = &sb[(size_t)cc * 6];
                for (int64_t c2 = 0; c2 < nc; ++c2) {
                    const double *r2 = &sb[(size_t)c2 * 6];
                    if (near(r2[2], -q[3]) && near(r2[3], -q[2]) && near(r2[4], q[4]) && near(r2[5], q[5])) {
                        rmir[(size_t)cc] = (int)c2;
                        break;
                    }
                }
                ok = rmir[(size_t)cc] >= 0;
            }
            for (int64_t cc = 0; cc < nc && ok; ++cc) ok = rmir[(size_t)rmir[(size_t)cc]] == (int)cc;
            if (ok) {
                for (int64_t cc = 0; cc < nc; ++cc) {
                    const int m2 = rmir[(size_t)cc];
                    if (m2 < cc) continue;  // (listed with its partner)
                    itc.push_back((int)cc);
                    itc2.push_back(m2 == cc ? -1 : m2);
                }
                h.wmirror = true;
            }
        }
        h.witems = h.wmirror ? (int)itc.size() : (int)nc;
        const int64_t ni = h.witems;
        // (GRAVHMC_LW_LDS_PAD: extra LDS per workgroup in KB -- a diagnostic that lowers the workgroups per CU)
        h.wlds = lonsymw_lds_doubles((int)n, h.nf) * sizeof(double) + (size_t)env_int("GRAVHMC_LW_LDS_PAD", 0) * 1024;
        HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(lonsymw_sweep_fn(h.nf)), h.wlds));
        h.wgrid = (int)std::min<int64_t>(ni, (int64_t)c->cus * 8);
        // parts of the forward product: ~8 waves per SIMD over the chip, at least 8 rows of T^ per part
        const int64_t waves_row = ((int64_t)na * h.nfp + 63) / 64;
        int64_t parts = std::max<int64_t>(1, ((int64_t)c->cus * env_int("GRAVHMC_LW_WAVES_PER_CU", 32) + waves_row - 1) / waves_row);
        parts = std::min<int64_t>(parts, std::max<int64_t>(1, ni / 8));
        parts = std::min<int64_t>(parts, 64);
        h.wrows = (int)((ni + parts - 1) / parts);
        h.wparts = (int)((ni + h.wrows - 1) / h.wrows);
        h.wfwd = env_int("GRAVHMC_LW_FWD", 0);
        h.wbreak = env_int("GRAVHMC_LW_BREAK", 0);  // (diagnostic: phases of the sweep switched off -- wrong results, timing only)
        TRY(dalloc(c, &h.tw, (size_t)n, false));
        TRY(dalloc(c, &h.That, (size_t)ni * (size_t)na * (size_t)h.nfp, false));
        TRY(dalloc(c, &h.Rhat, (size_t)na * (size_t)h.nfp));
        TRY(dalloc(c, &h.Xhat, (size_t)nc * (size_t)h.nf));
        TRY(dalloc(c, &h.Dpart, (size_t)h.wparts * (h.wmirror ? 2 : 1) * (size_t)na * (size_t)h.nfp));
        if (h.wmirror) {
            TRY(up(&h.item_c, itc));
            TRY(up(&h.item_c2, itc2));
            TRY(up(&h.amir, amirv));
        }
        lonsymh_twiddle_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream>>>((int)n, h.tw);
        lonsymh_table_kernel<<<dim3((unsigned)(ni * na)), dim3(64), 0, c->stream>>>(h.T, h.ldT, (int)n, h.nf, (int)na, h.tw, h.That, h.nfp,
                                                                                     h.wmirror ? h.item_c : nullptr);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
        h.wide = true;
    }
    if (!h.harm && !h.wide && !h.direct_ok) return no(direct_why);
    h.on = true;
    return GH_OK;
}

static LonWideGeom lonsymw_geom(const gh_ctx *c)
{
    const LonSymHost &h = *c->ls;
    LonWideGeom g;
    g.n = h.n;
    g.nf = h.nf;
    g.na = h.na;
    g.nc = h.nc;
    g.nfp = h.nfp;
    g.brk = h.wbreak;
    g.nitems = h.witems;
    g.planes = h.wmirror ? 2 : 1;
    g.item_c = h.wmirror ? h.item_c : nullptr;
    g.item_c2 = h.wmirror ? h.item_c2 : nullptr;
    g.amir = h.wmirror ? h.amir : nullptr;
    g.parts = h.wparts;
    g.rows_per_part = h.wrows;
    g.That = h.That;
    g.tw = h.tw;
    g.Rhat = h.Rhat;
    g.Xhat = h.Xhat;
    g.Dpart = h.Dpart;
    g.slot_first = h.slot_first;
    g.slot_x = h.slot_x;
    g.xptr = h.xptr;
    g.xobs = h.xobs;
    g.N = c->N;
    return g;
}

static LonHarmGeom lonsymh_geom(const gh_ctx *c)
{
    const LonSymHost &h = *c->ls;
    LonHarmGeom g;
    g.n = h.n;
    g.nf = h.nf;
    g.na = h.na;
    g.nc = h.nc;
    g.That = h.That;
    g.tw = h.tw;
    g.Rhat = h.Rhat;
    g.Dpart = h.Dpart;
    g.slot_first = h.slot_first;
    g.slot_x = h.slot_x;
    g.n_xslots = h.n_xslots;
    g.xslot = h.xslot;
    g.xptr = h.xptr;
    g.xobs = h.xobs;
    g.N = c->N;
    g.dbg = h.dbg;
    return g;
}

static bool lonsym_harmonic(const gh_ctx *c) { return c->ls && c->ls->on && c->ls->harm; }

static bool lonsym_on(const gh_ctx *c) { return c->ls && c->ls->on; }
// (harmonic forms: the pass delivers ONE finished slab row and the classes' sums)
static bool lonsym_one_row(const gh_ctx *c) { return c->ls && c->ls->on && (c->ls->harm || c->ls->wide); }
static int lonsym_classes(const gh_ctx *c) { return c->ls->na; }
// (workgroups of the pass = rows of the partial sums of p'p the trajectory code reads back)
static int lonsym_grid(const gh_ctx *c) { return c->ls->harm ? c->ls->hgrid : c->ls->wide ? c->ls->wgrid : c->ls->grid; }
static int64_t lonsym_table_bytes(const gh_ctx *c) { return c->ls->ldT * c->ls->nc * (int64_t)sizeof(double); }

static int launch_lonsym(gh_ctx *c, SweepArgs &a)
{
    const LonSymHost &h = *c->ls;
    a.ld = c->ld;
    a.M = c->M;
    if (h.harm) {
        // harmonic domain: R^ in front of the pass, the finished slab row (and the classes' sums) behind it
        const LonHarmGeom g = lonsymh_geom(c);
        LonSymHost &hw = *c->ls;
        // (R^ of this very residual vector may have come with the fused epilogue that produced it)
        if ((a.mode & SW_ADJ) && !(h.fused && a.r && h.rhat_of == a.r))
            lonsymh_rhat_kernel<<<dim3((unsigned)h.na), dim3(256), 0, c->stream>>>(g, a.r);
        hipLaunchKernelGGL(lonsymh_fn(h.rw), dim3((unsigned)h.hgrid), dim3(LH_THREADS), h.hlds, c->stream, g, a,
                           c->weighted ? c->wm : nullptr);
        if (a.mode & SW_FWD) {
            if (h.fused) {
                hw.post_pending = true;  // finalize() turns the partials into d, r and R^ in one launch
                hw.post_slab = a.slab;
                hw.post_dsum = a.dsum;
            } else {
                lonsymh_post_kernel<<<dim3((unsigned)h.na), dim3(512), 0, c->stream>>>(g, h.hgrid, c->ld, a.slab, a.dsum);
            }
        }
        return GH_OK;
    }
    if (h.wide) {
        // streamed harmonic form: R^ in front, the forward product over ranges of cell rows and the classes' inverse
        // transforms behind the row-parallel pass
        const LonWideGeom g = lonsymw_geom(c);
        if (a.mode & SW_ADJ) lonsymw_rhat_kernel<<<dim3((unsigned)h.na), dim3(LW_THREADS), 0, c->stream>>>(g, a.r);
        hipLaunchKernelGGL(lonsymw_sweep_fn(h.nf), dim3((unsigned)h.wgrid), dim3(LW_THREADS), h.wlds, c->stream, g, a,
                           c->weighted ? c->wm : nullptr);
        if (a.mode & SW_FWD) {
            const int64_t tot = (int64_t)h.na * h.nfp;
            const dim3 fgrid((unsigned)((tot + LW_THREADS - 1) / LW_THREADS), (unsigned)h.wparts);
            switch (h.wfwd) {  // (GRAVHMC_LW_FWD: rows in flight per thread / non-temporal loads; tuning, same results)
            case 1: lonsymw_forward_kernel<8, true><<<fgrid, dim3(LW_THREADS), 0, c->stream>>>(g); break;
            case 2: lonsymw_forward_kernel<16, false><<<fgrid, dim3(LW_THREADS), 0, c->stream>>>(g); break;
            case 3: lonsymw_forward_kernel<16, true><<<fgrid, dim3(LW_THREADS), 0, c->stream>>>(g); break;
            case 4: lonsymw_forward_kernel<4, false><<<fgrid, dim3(LW_THREADS), 0, c->stream>>>(g); break;
            default: lonsymw_forward_kernel<8, false><<<fgrid, dim3(LW_THREADS), 0, c->stream>>>(g); break;
            }
            lonsymw_post_kernel<<<dim3((unsigned)h.na), dim3(LW_THREADS), 0, c->stream>>>(g, c->ld, a.slab, a.dsum);
        }
        return GH_OK;
    }
    hipLaunchKernelGGL(lonsym_fn(h.items, h.W, h.thr), dim3((unsigned)h.grid), dim3((unsigned)h.thr), h.lds, c->stream, lonsym_geom(c), a,
                       c->weighted ? c->wm : nullptr);
    return GH_OK;
}

// The sweep's D^ partials as one finished slab row (callers that do not go through the fused epilogue:
// gh_forward, the generic epilogues).
static int lonsym_post_now(gh_ctx *c)
{
    if (!lonsym_harmonic(c) || !c->ls->post_pending) return GH_OK;
    LonSymHost &h = *c->ls;
    lonsymh_post_kernel<<<dim3((unsigned)h.na), dim3(512), 0, c->stream>>>(lonsymh_geom(c), h.hgrid, c->ld, h.post_slab, h.post_dsum);
    HIPCHK(c, hipGetLastError());
    h.post_pending = false;
    return GH_OK;
}

// after a synchronisation point: did a fused epilogue give up waiting for its class blocks?
static int lonsym_epilogue_check(gh_ctx *c)
{
    if (!lonsym_harmonic(c) || !c->ls->fused || !c->ls->epi_abort) return GH_OK;
    unsigned w[4] = {0, 0, 0, 0};
    HIPCHK(c, hipMemcpyAsync(w, c->ls->epi_abort, sizeof w, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (w[0] == 0u) return GH_OK;
    HIPCHK(c, hipMemsetAsync(c->ls->epi_abort, 0, sizeof w, c->stream));
    c->ls->fused = false;  // (the three-launch epilogue from here on)
    c->ls->rhat_of = nullptr;
    return fail(c, GH_ERR_HIP, "shift-invariant store: the one-launch epilogue timed out waiting for its class blocks (GPU shared?); "
                               "the evaluation is void, the three-launch form is used from here on");
}
