// libgravhmc host side: the per-evaluation epilogue (slab -> d, regulariser, residual, scalars)
// and the work buffers.  Included once by gravhmc.hip.
#pragma once

// slab -> d ; regulariser ; residual + scalars.  x: position the forward belongs to.
// slab (c->grid rows) -> d_out (+ per-block partial sums of d + grav_fix).  Many slab rows
// (small problems spread over many workgroups) are summed in two passes so that no thread walks
// hundreds of rows serially.
static void reduce_slab(gh_ctx *c, const double *gfix, double *d_out)
{
    (void)lonsym_post_now(c);  // (harmonic shift-invariant store: the sweep left D^ partials, not a slab row)
    const int rows = c->slab_live > 0 ? c->slab_live : c->grid;
    if (rows > 64 && c->slab2) {
        const int nseg = c->slab2_rows;
        reduce_slab_kernel<<<dim3(c->n_dpart, nseg), dim3(32, 8), 0, c->stream>>>(c->slab, rows, c->ld, c->N,
                                                                                  nullptr, c->slab2, c->dpart);
        reduce_slab_kernel<<<dim3(c->n_dpart, 1), dim3(32, 8), 0, c->stream>>>(c->slab2, nseg, c->ld, c->N,
                                                                               gfix, d_out, c->dpart);
    } else {
        reduce_slab_kernel<<<dim3(c->n_dpart, 1), dim3(32, 8), 0, c->stream>>>(c->slab, rows, c->ld, c->N, gfix,
                                                                               d_out, c->dpart);
    }
}

static int finalize(gh_ctx *c, const double *x, const gh_ctx::StateSet &o)
{
    double *d_out = o.d, *r_out = o.r, *greg_out = o.greg, *scal_out = o.scal;
    RegArgs ra{};
    ra.ms_grad_den_mw = 0;
    ra.kind = c->reg_kind;
    ra.M = c->M;
    ra.nz = c->shape[0];
    ra.ny = c->shape[1];
    ra.nx = c->shape[2];
    ra.alpha = c->alpha;
    ra.beta = c->beta;
    ra.x = x;
    ra.mwapr = c->mwapr;
    ra.wm2 = c->wm2;
    ra.greg = greg_out;
    ra.regpart = c->regpart;
    const double *gfix = c->have_fix ? c->gfix : nullptr;
    const double *regpart = c->regpart;
    int n_regpart = c->n_regpart;
    const double *src;
    int nseg;
    // harmonic shift-invariant store: the sweep left D^ partials; either the fused epilogue below consumes them or
    // they are turned into the slab row first
    const bool fused_epi = lonsym_harmonic(c) && c->ls->fused && c->ls->post_pending && o.part != nullptr;
    if (!fused_epi) TRY(lonsym_post_now(c));
    if (shard_rows(c)) {
        // observations sharded over the ranks: d, r of the local rows; the mean and |r|^2 are sums over all
        // ranks (two scalar all-reduces); the regulariser is replicated with the model
        reg_kernel<<<dim3(c->n_regpart), dim3(256), 0, c->stream>>>(ra);
        RowsFinishArgs fa{};
        fa.N = c->N;
        fa.ld = c->ld;
        fa.N_global = c->sh.N_global;
        fa.nseg = c->slab_live > 0 ? c->slab_live : c->grid;
        if (c->wv.on) {
            // the local rows of the compressed operator (every rank compressed its own rows of Aw): one finished row
            TRY(wavelet_forward(c, x, c->slab));
            fa.nseg = 1;
        }
        fa.n_regpart = c->n_regpart;
        fa.src = c->slab;
        fa.gfix = gfix;
        fa.dobs_c = c->dobs_c;
        fa.regpart = c->regpart;
        fa.alpha = c->alpha;
        fa.d = d_out;
        fa.r = r_out;
        fa.scal = scal_out;
        fa.rbuf = c->sh.rbuf;
        rows_stage_a_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(fa);
        TRY(comm_allreduce(c, c->sh.rbuf, 1));
        rows_stage_b_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(fa);
        TRY(comm_allreduce(c, c->sh.rbuf + 1, 1));
        rows_stage_c_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(fa);
        HIPCHK(c, hipGetLastError());
        o.pending = false;
        return GH_OK;
    }
    if (c->sh.kind != 0) {
        // sharded cells: local forward partial and local regulariser sum travel in ONE
        // all-reduce, then every rank finishes the (replicated) data part identically
        double *buf = c->sh.buf;
        reduce_slab(c, nullptr, buf);
        if (c->sh.halo) {
            // stencil regulariser: the boundary planes of x travel with the forward partial, the
            // regulariser (which needs them) is summed by a second, two-double all-reduce
            gh_ctx::Shard &sh = c->sh;
            const int64_t P = sh.P, nh = 2 * (int64_t)sh.world * P;
            double *hb = buf + c->ld + 8;
            halo_pack_kernel<<<dim3((unsigned)std::min<int64_t>(1024, (nh + 255) / 256)), dim3(256), 0, c->stream>>>(
                x, c->M, P, sh.rank, sh.world, hb);
            TRY(comm_allreduce(c, buf, (int64_t)c->ld + 8 + nh));
            ra.nz = c->shape[0];
            ra.k0 = sh.m0 / P;
            ra.xlo = sh.rank > 0 ? hb + ((int64_t)(sh.rank - 1) * 2 + 1) * P : nullptr;
            ra.xhi = sh.rank + 1 < sh.world ? hb + (int64_t)(sh.rank + 1) * 2 * P : nullptr;
            ra.alo = sh.alo;
            ra.ahi = sh.ahi;
            reg_kernel<<<dim3(c->n_regpart), dim3(256), 0, c->stream>>>(ra);
            sum_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(c->regpart, c->n_regpart, sh.rb);
            TRY(comm_allreduce(c, sh.rb, 2));
            regpart = sh.rb;
        } else {
            reg_kernel<<<dim3(c->n_regpart), dim3(256), 0, c->stream>>>(ra);
            sum_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(c->regpart, c->n_regpart, buf + c->ld);
            TRY(comm_allreduce(c, buf, c->ld + 2));
            regpart = buf + c->ld;
        }
        src = buf;
        nseg = 1;
        n_regpart = 1;
    } else if (c->wv.on) {
        // forward through the compressed operator: d_out is already complete
        TRY(wavelet_forward(c, x, d_out));
        reg_kernel<<<dim3(c->n_regpart), dim3(256), 0, c->stream>>>(ra);
        src = d_out;
        nseg = 1;
    } else if (fused_epi) {
        // harmonic shift-invariant store: partial sums -> d -> mean -> r -> R^ of the next adjoint, and the
        // regulariser, in ONE launch (lonsymh_epilogue_kernel)
        LonSymHost &h = *c->ls;
        LhEpiArgs e{};
        e.nparts = h.hgrid;
        e.n_dpart = c->n_dpart;
        e.ld = c->ld;
        e.N = c->N;
        e.gfix = gfix;
        e.dobs_c = c->dobs_c;
        e.d = d_out;
        e.r = r_out;
        e.scal = scal_out;
        e.r2part = o.part;
        e.csum = h.csum;
        h.epi_tag += 1u;
        if (h.epi_tag > 0xf0000000u) {
            HIPCHK(c, hipMemsetAsync(h.csum, 0, sizeof(unsigned long long) * 2 * (size_t)h.na, c->stream));
            h.epi_tag = 1u;
        }
        e.tag = h.epi_tag;
        e.abort_w = h.epi_abort;
        e.ra = ra;
        e.ra.regpart = o.part + c->n_dpart;
        lonsymh_epilogue_kernel<<<dim3((unsigned)(h.na + c->n_regpart)), dim3(256), 0, c->stream>>>(lonsymh_geom(c), e);
        HIPCHK(c, hipGetLastError());
        h.post_pending = false;
        h.rhat_of = r_out;
        o.pending = true;
        return GH_OK;
    } else if (c->dsum_live && o.part) {
        // the slab rows' sums at hand (and N >= 2048): the whole epilogue in one launch
        ReduceFinishArgs fa{};
        fa.slab = c->slab;
        fa.n_rows_slab = c->slab_live > 0 ? c->slab_live : c->grid;
        fa.n_dpart = c->n_dpart;
        fa.n_regpart = c->n_regpart;
        fa.ld = c->ld;
        fa.N = c->N;
        fa.dsum = c->dsum;
        fa.n_dsum = c->dsum_n > 0 ? c->dsum_n : fa.n_rows_slab;
        fa.gfix_sum = gfix ? c->gfix_sum : 0.0;
        fa.gfix = gfix;
        fa.dobs_c = c->dobs_c;
        fa.d = d_out;
        fa.r = r_out;
        fa.scal = scal_out;
        fa.r2part = o.part;
        fa.ra = ra;
        fa.ra.regpart = o.part + c->n_dpart;
        reduce_finish_kernel<<<dim3((unsigned)(c->n_dpart + c->n_regpart)), dim3(256), 0, c->stream>>>(fa);
        HIPCHK(c, hipGetLastError());
        o.pending = true;
        return GH_OK;
    } else if (c->grid > 64 && c->slab2) {
        // many slab rows: first stage of the reduction and the regulariser share one launch,
        // finish_kernel sums the 16 segments
        nseg = c->slab2_rows;
        reduce_reg_kernel<<<dim3((unsigned)(c->n_dpart * nseg + c->n_regpart)), dim3(256), 0, c->stream>>>(
            c->slab, c->slab_live > 0 ? c->slab_live : c->grid, c->ld, nseg, c->n_dpart, c->slab2, ra);
        src = c->slab2;
    } else {
        reg_kernel<<<dim3(c->n_regpart), dim3(256), 0, c->stream>>>(ra);
        src = c->slab;
        nseg = c->slab_live > 0 ? c->slab_live : c->grid;
    }
    o.pending = false;
    FinishArgs fa;
    fa.N = c->N;
    fa.ld = c->ld;
    fa.nseg = nseg;
    fa.n_regpart = n_regpart;
    fa.src = src;
    fa.gfix = gfix;
    fa.dobs_c = c->dobs_c;
    fa.regpart = regpart;
    fa.alpha = c->alpha;
    fa.d = d_out;
    fa.r = r_out;
    fa.scal = scal_out;
    finish_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(fa);
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

// scal[0..2] of a state set, summed from the partials of a one-launch epilogue if still pending
static int scal_ready(gh_ctx *c, const gh_ctx::StateSet &o)
{
    if (!o.pending) return GH_OK;
    scal_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(o.part, c->n_dpart, c->n_regpart, c->alpha, o.scal);
    HIPCHK(c, hipGetLastError());
    o.pending = false;
    return GH_OK;
}

// forward sweep of x (device) + finalize
static int eval_forward(gh_ctx *c, const double *x, const gh_ctx::StateSet &o)
{
    if (!c->wv.on) {
        SweepArgs a{};
        a.mode = SW_FWD;
        a.x_in = x;
        a.slab = c->slab;
        TRY(launch_sweep(c, a));
    }
    return finalize(c, x, o);
}

static int ensure_work(gh_ctx *c)
{
    const size_t M = (size_t)c->M, ld = (size_t)c->ld;
    TRY(dalloc(c, &c->scal_all, 16));
    for (int i = 0; i < 4; ++i) {
        TRY(dalloc(c, &c->st[i].r, ld));
        TRY(dalloc(c, &c->st[i].greg, M));
        TRY(dalloc(c, &c->st[i].d, ld));
        c->st[i].scal = c->scal_all + 4 * i;
        TRY(dalloc(c, &c->xb[i], M));
    }
    TRY(dalloc(c, &c->pb[0], M));
    TRY(dalloc(c, &c->pb[1], M));
    TRY(dalloc(c, &c->pn, M));
    if (c->n_panels > 1 || shard_rows(c)) TRY(dalloc(c, &c->gbuf, M));
    // (N > 16384: the team sweep writes one slab row per team, up to 128, whatever the panel grid)
    // (the harmonic forms of the shift-invariant store deliver ONE finished slab row)
    TRY(dalloc(c, &c->slab, (size_t)(lonsym_one_row(c) ? 1 : c->n_panels > 1 ? std::max(c->grid, 128) : c->grid) * ld));
    if (c->grid > 64) {
        // segments left for the single-block finish_kernel: as many as keep its read at ~128 KB
        // (C1: 16 x 608 rows; C2: 1 x 10^4 -- sixteen there made that one block read 1.3 MB, 57 us)
        c->slab2_rows = (int)std::max<int64_t>(1, std::min<int64_t>(16, 16384 / (int64_t)ld));
        TRY(dalloc(c, &c->slab2, (size_t)c->slab2_rows * ld));
    }
    c->n_dpart = (int)((c->ld + 31) / 32);
    if (c->ld >= 2048 && ((c->TW > 1 && c->n_panels == 1 && !c->mf) || lonsym_on(c)) && env_int("GRAVHMC_EPILOGUE1", 1) != 0) {
        TRY(dalloc(c, &c->dsum, (size_t)std::max(std::max(c->grid, 128), lonsym_on(c) ? lonsym_classes(c) : 0)));
        for (int i = 0; i < 4; ++i) TRY(dalloc(c, &c->st[i].part, (size_t)c->n_dpart + (size_t)((c->M + 255) / 256)));
    }
    c->n_regpart = (int)((c->M + 255) / 256);
    c->n_pp0 = (int)std::min<int64_t>(1024, (c->M + 255) / 256);
    TRY(dalloc(c, &c->dpart, (size_t)c->n_dpart));
    TRY(dalloc(c, &c->regpart, (size_t)c->n_regpart));
    TRY(dalloc(c, &c->pp_part, (size_t)c->n_teams));
    TRY(dalloc(c, &c->ppn_part, (size_t)c->n_teams));
    TRY(dalloc(c, &c->pp0_part, (size_t)c->n_pp0));
    TRY(dalloc(c, &c->pn0_part, (size_t)c->n_pp0));
    TRY(dalloc(c, &c->tmpM, M));
    TRY(dalloc(c, &c->tmpN, ld));
    TRY(dalloc(c, &c->low, M));
    TRY(dalloc(c, &c->high, M));
    if (!c->mwapr) {
        TRY(dalloc(c, &c->mwapr, M));
    }
    if (!c->wm2) {
        TRY(dalloc(c, &c->wm2, M));
    }
    if (!c->h_scal) {
        c->h_scal_n = 16 + 2 * (size_t)c->n_teams + 2 * (size_t)c->n_pp0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_scal, c->h_scal_n * sizeof(double)));
    }
    return GH_OK;
}

static int need(gh_ctx *c, bool cond, const char *what)
{
    if (!cond) return fail(c, GH_ERR_ARG, "%s", what);
    return GH_OK;
}
