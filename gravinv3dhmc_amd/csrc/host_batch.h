// libgravhmc host side: buffers and evaluation of the fp64-MFMA chain batch (batch.hip.h).
// Included once by gravhmc.hip.
#pragma once

// --------------------------------------------------------------- batched chains (MFMA)

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
    // forward: 512-row blocks x column blocks, about 4 workgroups per CU in total
    const int rowblocks = (int)((c->ld + 511) / 512);
    int colblocks = std::max(1, (c->cus * 4 + rowblocks - 1) / rowblocks);
    int64_t cpb = (c->M + colblocks - 1) / colblocks;
    cpb = (cpb + 15) / 16 * 16;
    b.cols_per_block = cpb;
    b.n_colblocks = (int)((c->M + cpb - 1) / cpb);
    TRY(dalloc(c, &b.slab, (size_t)b.n_colblocks * L16));
    b.n_regblocks = (int)((c->M + 15) / 16);
    TRY(dalloc(c, &b.regpart, (size_t)b.n_regblocks * CB));
    const int64_t ntiles = (c->M + 15) / 16;
    const int64_t npairs = (ntiles + 1) / 2;  // a wave owns two adjacent column tiles
    const int wgs = (int)std::min<int64_t>((npairs + 3) / 4, (int64_t)c->cus * 4);
    b.n_waves = wgs * 4;
    TRY(dalloc(c, &b.pp_part, (size_t)b.n_waves * CB));
    b.n_pp0 = (int)std::min<int64_t>(512, (c->M + 15) / 16);
    TRY(dalloc(c, &b.pp0_part, (size_t)b.n_pp0 * CB));
    HIPCHK(c, hipHostMalloc((void **)&b.h, sizeof(double) * (size_t)(CB * 4 + (b.n_waves + 2 * b.n_pp0) * CB)));
    // the adjoint GEMM wants G in MFMA operand order; 288 GB of HBM usually has room for the
    // second copy (C2: 40 GB + 40 GB).  Without it the kernel reads the column-major matrix.
    if (env_int("GRAVHMC_BATCH_RELAYOUT", 1)) {
        size_t free_b = 0, total_b = 0;
        const size_t need_b = sizeof(double) * (size_t)ntiles * 16 * (size_t)c->ld;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b > need_b + ((size_t)2 << 30)) {
            void *ptr = nullptr;
            if (hipMalloc(&ptr, need_b) == hipSuccess) {
                c->allocs.push_back(ptr);
                b.Gb = static_cast<double *>(ptr);
                batch_relayout_kernel<<<dim3(1 << 16), dim3(256), 0, c->stream>>>(c->G, c->ld, c->M, (int)(c->ld / 16),
                                                                                  ntiles, b.Gb);
                HIPCHK(c, hipGetLastError());
            } else {
                (void)hipGetLastError();
            }
        }
    }
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

// forward of all chains at X, then regulariser and residuals into (D, GREG, Rt, scal)
static int batch_evaluate(gh_ctx *c, const double *X, double *D, double *GREG, double *Rt, double *scal = nullptr)
{
    gh_ctx::Batch &b = c->bt;
    BatchFwdArgs f;
    f.G = c->G;
    f.ld = c->ld;
    f.M = c->M;
    f.N = c->N;
    f.X = X;
    f.cols_per_block = b.cols_per_block;
    f.slab = b.slab;
    bool timed;
    TRY(batch_time_begin(c, timed));
    batch_forward_kernel<<<dim3((unsigned)((c->ld + 511) / 512), (unsigned)b.n_colblocks), dim3(256), 0,
                           c->stream>>>(f);
    TRY(batch_time_end(c, timed));
    const int64_t n16 = c->ld * CB;
    batch_reduce_kernel<<<dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream>>>(b.slab, b.n_colblocks,
                                                                                        n16, D);
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
