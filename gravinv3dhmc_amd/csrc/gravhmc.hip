// libgravhmc: host side of the C-ABI declared in include/gravhmc.h (HIP, gfx950 only).
// One context = one GPU + one stream + one inversion problem resident in HBM.
#include "../../include/gravhmc.h"
#include "kernels.hip.h"
#include "batch.hip.h"
#include "resident.hip.h"
#include "resbatch.hip.h"
#include "teamsweep.hip.h"
#include "mfbatch.hip.h"
#include "batchteam.hip.h"
#include "lonsym.hip.h"
#include "lonsymh.hip.h"
#include "lonres.hip.h"
#include "lonsymw.hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <thread>
#include <vector>

using namespace ghk;

#include "host_ctx.h"
#include "host_sweep.h"
#include "host_lonsym.h"
#include "host_comm.h"
#include "host_wavelet.h"
#include "host_eval.h"
#include "host_resident.h"
#include "host_resbatch.h"
#include "host_lonres.h"
#include "host_batch.h"

// ------------------------------------------------------------------------- C-ABI

extern "C" {

const char *gh_last_error(const gh_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int gh_create(gh_ctx **out, int device, int64_t N, int64_t M)
{
    if (!out || N <= 0 || M <= 0) return fail(nullptr, GH_ERR_ARG, "gh_create: bad arguments");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(nullptr, GH_ERR_HIP, "gh_create: no HIP device available (%s); libgravhmc has no CPU path",
                    hipGetErrorString(e));
    if (device < 0 || device >= ndev)
        return fail(nullptr, GH_ERR_ARG, "gh_create: device %d out of range (%d devices)", device, ndev);
    gh_ctx *c = new gh_ctx();
    c->device = device;
    c->N = N;
    c->M = M;
    c->ld = (N + 15) / 16 * 16;
    hipDeviceProp_t prop;
    if ((e = hipSetDevice(device)) != hipSuccess || (e = hipGetDeviceProperties(&prop, device)) != hipSuccess ||
        (e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) {
        int rc = fail(nullptr, GH_ERR_HIP, "gh_create: %s", hipGetErrorString(e));
        delete c;
        return rc;
    }
    c->cus = prop.multiProcessorCount;
    int rc = configure_sweep(c);
    if (rc != GH_OK) {
        g_create_error = c->err;
        hipStreamDestroy(c->stream);
        delete c;
        return rc;
    }
    *out = c;
    return GH_OK;
}

void gh_destroy(gh_ctx *c)
{
    if (!c) return;
    for (gh_ctx *k : c->kids) gh_destroy(k);
    c->kids.clear();
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->sh.comm) {
        std::string err;
        RcclApi *api = rccl_api(err);
        if (api && api->CommDestroy) api->CommDestroy(c->sh.comm);
    }
    if (c->sh.hbuf) hipHostFree(c->sh.hbuf);
    if (c->bt.h) hipHostFree(c->bt.h);
    if (c->rs.ls.h_stage) hipHostFree(c->rs.ls.h_stage);
    if (c->rs.ls.h_xstage) hipHostFree(c->rs.ls.h_xstage);
    if (c->ls) {
        if (c->ls->res.h_stage) hipHostFree(c->ls->res.h_stage);
        if (c->ls->res.ev0) hipEventDestroy(c->ls->res.ev0);
        if (c->ls->res.ev1) hipEventDestroy(c->ls->res.ev1);
    }
    for (gh_ctx::Pinned &pm : c->pinned) hipHostFree(pm.base);
    for (void *p : c->allocs) hipFree(p);
    if (c->h_scal) hipHostFree(c->h_scal);
    for (hipEvent_t ev : c->ev) hipEventDestroy(ev);
    if (c->copy_ev) hipEventDestroy(c->copy_ev);
    if (c->copy_stream) hipStreamDestroy(c->copy_stream);
    if (c->rs.ev0) hipEventDestroy(c->rs.ev0);
    if (c->rs.ev1) hipEventDestroy(c->rs.ev1);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c->ls;
    delete c;
}

int gh_device_info(const gh_ctx *c, char *name256, int *cus, int64_t *mem_bytes)
{
    if (!c) return GH_ERR_ARG;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, c->device) != hipSuccess) return GH_ERR_HIP;
    if (name256) {
        // some ROCm builds leave prop.name empty: fall back to the ISA name
        snprintf(name256, 256, "%s%s%s", prop.name, prop.name[0] ? " " : "AMD Instinct ", prop.gcnArchName);
    }
    if (cus) *cus = prop.multiProcessorCount;
    if (mem_bytes) *mem_bytes = (int64_t)prop.totalGlobalMem;
    return GH_OK;
}

int gh_synchronize(gh_ctx *c)
{
    if (!c) return GH_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

int gh_set_obs(gh_ctx *c, const double *a, const double *b, const double *cc)
{
    if (!c || !a || !b || !cc) return fail(c, GH_ERR_ARG, "gh_set_obs: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    const double *src[3] = {a, b, cc};
    for (int i = 0; i < 3; ++i) {
        TRY(dalloc(c, &c->obs[i], (size_t)c->N));
        TRY(h2d(c, c->obs[i], src[i], (size_t)c->N));
    }
    // (tesseroids: one height for all observations lets the matrix-free batch hoist what depends on
    // the radius alone out of the entries, mfbatch.hip.h)
    c->obs_h_uniform = true;
    for (int64_t i = 1; i < c->N && c->obs_h_uniform; ++i) c->obs_h_uniform = cc[i] == cc[0];
    c->obs_h0 = cc[0];
    c->have_obs = true;
    return GH_OK;
}

int gh_set_cells(gh_ctx *c, const double *bounds6, int kind, double ratio)
{
    if (!c || !bounds6) return fail(c, GH_ERR_ARG, "gh_set_cells: null pointer");
    if (kind != GH_CELL_PRISM && kind != GH_CELL_TESSEROID)
        return fail(c, GH_ERR_ARG, "gh_set_cells: kind must be 0 (prism) or 1 (tesseroid)");
    if (kind == GH_CELL_TESSEROID && !(ratio > 0))
        return fail(c, GH_ERR_ARG, "Invalid ratio %g. Must be > 0.", ratio);
    HIPCHK(c, hipSetDevice(c->device));
    TRY(dalloc(c, &c->bounds, (size_t)c->M * 6));
    TRY(h2d(c, c->bounds, bounds6, (size_t)c->M * 6));
    c->cell_kind = kind;
    c->ratio = ratio;
    c->have_cells = true;
    return GH_OK;
}

int gh_set_matrix_free(gh_ctx *c, int enable)
{
    if (!c) return GH_ERR_ARG;
    if (c->have_G || c->slab) return fail(c, GH_ERR_ARG, "gh_set_matrix_free: call before gh_build_G");
    if (c->ls) {
        c->mf_before_ls = enable != 0;  // (takes effect when the shift-invariant store is switched off)
        return GH_OK;
    }
    c->mf = enable != 0;  // (the passes are partitioned in gh_build_G, once the cell kind is known)
    return GH_OK;
}

int gh_set_shift_invariant(gh_ctx *c, int enable)
{
    if (!c) return GH_ERR_ARG;
    if (c->have_G || c->slab) return fail(c, GH_ERR_ARG, "gh_set_shift_invariant: call before gh_build_G");
    // (the store is a flavour of the matrix-free mode -- G is never stored -- so enabling it sets c->mf;
    // disabling it puts c->mf back to what gh_set_matrix_free last asked for)
    if (c->ls) c->mf = c->mf_before_ls;
    delete c->ls;
    c->ls = nullptr;
    if (enable) {
        c->ls = new LonSymHost();
        c->mf_before_ls = c->mf;
        c->mf = true;
    }
    return GH_OK;
}

int gh_shift_invariant_resident_stats(gh_ctx *c, int *workgroups, int64_t *launches, int64_t *evaluations, int64_t *trajectories,
                                      int *timeouts)
{
    if (!c) return GH_ERR_ARG;
    const bool on = lonsym_harmonic(c) && c->ls->res.state > 0;
    if (workgroups) *workgroups = on ? c->ls->hgrid : 0;
    if (launches) *launches = c->ls ? c->ls->res.launches : 0;
    if (evaluations) *evaluations = c->ls ? c->ls->res.evals : 0;
    if (trajectories) *trajectories = c->ls ? c->ls->res.trajectories : 0;
    if (timeouts) *timeouts = c->ls ? c->ls->res.aborts : 0;
    return GH_OK;
}

int gh_shift_invariant_info(const gh_ctx *c, int *n_lon, int *n_classes, int *n_rows, int64_t *table_bytes)
{
    if (!c) return GH_ERR_ARG;
    const bool on = lonsym_on(c);
    if (n_lon) *n_lon = on ? c->ls->n : 0;
    if (n_classes) *n_classes = on ? c->ls->na : 0;
    if (n_rows) *n_rows = on ? c->ls->nc : 0;
    if (table_bytes) *table_bytes = on ? lonsym_table_bytes(c) : 0;
    return GH_OK;
}

int gh_shift_invariant_harmonic(const gh_ctx *c, int *on, int *n_freq, int64_t *table_bytes, int *workgroups)
{
    if (!c) return GH_ERR_ARG;
    // on: 1 = the register form (lonsymh.hip.h), 2 = the streamed form for large grids (lonsymw.hip.h)
    const bool h = lonsym_one_row(c);
    if (on) *on = !h ? 0 : c->ls->harm ? 1 : 2;
    if (n_freq) *n_freq = h ? c->ls->nf : 0;
    if (table_bytes) *table_bytes = h ? (c->ls->harm ? (int64_t)c->ls->nc * c->ls->na * c->ls->nf : (int64_t)c->ls->witems * c->ls->na * c->ls->nfp) * 16 : 0;
    if (workgroups) *workgroups = h ? lonsym_grid(c) : 0;
    return GH_OK;
}

int gh_set_matrix_free_exact(gh_ctx *c, int exact)
{
    if (!c) return GH_ERR_ARG;
    if (c->have_G || c->slab) return fail(c, GH_ERR_ARG, "gh_set_matrix_free_exact: call before gh_build_G");
    c->mf_exact_req = exact != 0 ? 1 : 0;
    return GH_OK;
}

int gh_build_G(gh_ctx *c)
{
    if (!c) return GH_ERR_ARG;
    TRY(need(c, c->have_obs && c->have_cells, "gh_build_G: call gh_set_obs and gh_set_cells first"));
    HIPCHK(c, hipSetDevice(c->device));
    c->warn_cells = 0;
    c->leaves = 0;
    if (c->mf) {
        if (c->slab) return fail(c, GH_ERR_ARG, "gh_build_G: a matrix-free context is built once");
        c->mf_fused = c->ld <= 16384 && env_int("GRAVHMC_MF_FUSED", 1) != 0;
        if (c->cell_kind == GH_CELL_TESSEROID) {
            const int64_t N = c->N;
            TRY(dalloc(c, &c->tconv, (size_t)(6 * N)));
            tess_convert_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream>>>(
                c->obs[0], c->obs[1], c->obs[2], N, c->tconv, c->tconv + N, c->tconv + 2 * N,
                c->tconv + 3 * N, c->tconv + 4 * N, c->tconv + 5 * N);
            HIPCHK(c, hipGetLastError());
            c->mf_exact = c->mf_exact_req >= 0 ? c->mf_exact_req != 0 : env_int("GRAVHMC_MF_EXACT", 0) != 0;
            {
                // what depends on the cell alone, once per cell instead of once per (obs, cell) pair
                TRY(dalloc(c, &c->mf_cellc, (size_t)c->M * TESS_NC, false));
                tess_cellconst_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(
                    c->bounds, c->M, c->ratio, c->mf_cellc);
                HIPCHK(c, hipGetLastError());
                TRY(build_near_table(c));
            }
        }
        if (c->ls) TRY(lonsym_build(c));
        TRY(configure_mf(c));
        TRY(dalloc(c, &c->mf_stats, 1));
        c->have_G = true;
        c->weighted = false;
        c->chain_ready = false;
        c->bt.ready = false;
        return GH_OK;
    }
    if (!c->dense_ok)
        return fail(c, GH_ERR_UNSUPPORTED,
                    "N = %lld: more than 16384 observations per device: shard the observations or use "
                    "the matrix-free mode (gh_set_matrix_free)", (long long)c->N);
    TRY(dalloc(c, &c->G, (size_t)c->ld * (size_t)c->M, false));
    const int64_t total = c->ld * c->M;
    if (c->cell_kind == GH_CELL_PRISM) {
        const int64_t blocks = std::min<int64_t>((total + 255) / 256, 1 << 22);
        prism_gz_kernel<<<dim3((unsigned)blocks), dim3(256), 0, c->stream>>>(
            c->obs[0], c->obs[1], c->obs[2], c->bounds, c->N, c->M, c->ld, c->G);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    } else {
        double *conv = nullptr;
        int *err_cell = nullptr;
        TessStats *stats = nullptr;
        HIPCHK(c, hipMalloc((void **)&conv, sizeof(double) * 4 * (size_t)c->N));
        HIPCHK(c, hipMalloc((void **)&err_cell, sizeof(int) * (size_t)c->M));
        HIPCHK(c, hipMalloc((void **)&stats, sizeof(TessStats)));
        HIPCHK(c, hipMemsetAsync(err_cell, 0, sizeof(int) * (size_t)c->M, c->stream));
        HIPCHK(c, hipMemsetAsync(stats, 0, sizeof(TessStats), c->stream));
        const int64_t N = c->N;
        tess_convert_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream>>>(
            c->obs[0], c->obs[1], c->obs[2], N, conv, conv + N, conv + 2 * N, conv + 3 * N);
        const int64_t blocks = std::min<int64_t>((total + 63) / 64, 1 << 24);
        tess_gz_kernel<<<dim3((unsigned)blocks), dim3(64), 0, c->stream>>>(
            conv, conv + N, conv + 2 * N, conv + 3 * N, c->bounds, N, c->M, c->ld, c->ratio, c->G,
            err_cell, stats);
        HIPCHK(c, hipGetLastError());
        std::vector<int> herr((size_t)c->M);
        TessStats hs;
        HIPCHK(c, hipMemcpyAsync(herr.data(), err_cell, sizeof(int) * (size_t)c->M,
                                 hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(&hs, stats, sizeof hs, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        hipFree(conv);
        hipFree(err_cell);
        hipFree(stats);
        for (int v : herr)
            if (v != 0) c->warn_cells += 1;
        c->leaves = (int64_t)hs.leaves;
        if (hs.overflow) return fail(c, GH_ERR_OVERFLOW, "tesseroid stack overflow (> %d entries)", TESS_STACK);
    }
    c->have_G = true;
    c->weighted = false;
    c->chain_ready = false;
    c->bt.ready = false;
    return GH_OK;
}

int gh_kernel_stats(const gh_ctx *c, int64_t *warn_cells, int64_t *leaves)
{
    if (!c) return GH_ERR_ARG;
    if (warn_cells) *warn_cells = c->warn_cells;
    if (leaves) *leaves = c->leaves;
    return GH_OK;
}

int gh_upload_G(gh_ctx *c, const double *A, int64_t ld, int fortran_order)
{
    if (!c || !A) return fail(c, GH_ERR_ARG, "gh_upload_G: null pointer");
    if (ld < (fortran_order ? c->N : c->M)) return fail(c, GH_ERR_ARG, "gh_upload_G: ld too small");
    if (c->mf) return fail(c, GH_ERR_ARG, "gh_upload_G: context is matrix-free");
    if (!c->dense_ok) return fail(c, GH_ERR_UNSUPPORTED, "N = %lld: more than 16384 observations per device", (long long)c->N);
    HIPCHK(c, hipSetDevice(c->device));
    TRY(dalloc(c, &c->G, (size_t)c->ld * (size_t)c->M, false));
    HIPCHK(c, hipMemsetAsync(c->G, 0, sizeof(double) * (size_t)c->ld * (size_t)c->M, c->stream));
    if (fortran_order) {
        HIPCHK(c, hipMemcpy2DAsync(c->G, (size_t)c->ld * sizeof(double), A, (size_t)ld * sizeof(double),
                                   (size_t)c->N * sizeof(double), (size_t)c->M, hipMemcpyHostToDevice,
                                   c->stream));
    } else {
        // row-major N x M: transpose on the host in column panels (setup path, runs once)
        const int64_t panel = std::max<int64_t>(1, (int64_t)(64 << 20) / (int64_t)(c->N * sizeof(double)));
        std::vector<double> buf((size_t)std::min(panel, c->M) * (size_t)c->N);
        for (int64_t j0 = 0; j0 < c->M; j0 += panel) {
            const int64_t nb = std::min(panel, c->M - j0);
            for (int64_t i = 0; i < c->N; ++i)
                for (int64_t j = 0; j < nb; ++j) buf[(size_t)j * c->N + i] = A[i * ld + j0 + j];
            HIPCHK(c, hipMemcpy2DAsync(c->G + j0 * c->ld, (size_t)c->ld * sizeof(double), buf.data(),
                                       (size_t)c->N * sizeof(double), (size_t)c->N * sizeof(double),
                                       (size_t)nb, hipMemcpyHostToDevice, c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->have_G = true;
    c->weighted = false;
    c->chain_ready = false;
    c->bt.ready = false;
    return GH_OK;
}

int gh_download_G(gh_ctx *c, double *A, int64_t ld)
{
    if (!c || !A) return fail(c, GH_ERR_ARG, "gh_download_G: null pointer");
    TRY(need(c, c->have_G && !c->mf, "gh_download_G: no kernel matrix resident"));
    if (ld < c->N) return fail(c, GH_ERR_ARG, "gh_download_G: ld too small");
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy2DAsync(A, (size_t)ld * sizeof(double), c->G, (size_t)c->ld * sizeof(double),
                               (size_t)c->N * sizeof(double), (size_t)c->M, hipMemcpyDeviceToHost,
                               c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

int gh_weight(gh_ctx *c, double weightfactor, double *wm_out)
{
    if (!c) return GH_ERR_ARG;
    TRY(need(c, c->have_G, "gh_weight: no kernel matrix resident"));
    TRY(need(c, !c->weighted, "gh_weight: kernel is already weighted"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(dalloc(c, &c->wm, (size_t)c->M));
    TRY(dalloc(c, &c->wm2, (size_t)c->M));
    if (lonsym_on(c)) {
        lonsym_colnorm_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(
            lonsym_geom(c), c->ls->a_of, c->ls->m_of, weightfactor, c->wm);
    } else if (c->mf) {
        mf_colnorm_kernel<<<dim3((unsigned)((c->M + 3) / 4)), dim3(256), 0, c->stream>>>(mf_geom(c), weightfactor,
                                                                                        c->wm);
    } else if (shard_rows(c)) {
        // row blocks: a column's norm spans the ranks -- local sums of squares, all-reduce, then the power and
        // the scaling of the local rows
        const unsigned blocks = (unsigned)std::min<int64_t>(c->M, (int64_t)c->cus * 16);
        colnorm_kernel<<<dim3(blocks), dim3(256), 0, c->stream>>>(c->G, c->ld, c->M, 1.0, c->wm);
        HIPCHK(c, hipGetLastError());
        TRY(comm_allreduce(c, c->wm, c->M));
        std::vector<double> ss((size_t)c->M);
        TRY(d2h(c, ss.data(), c->wm, (size_t)c->M));
        for (auto &v : ss) v = (weightfactor == 0.5) ? std::sqrt(v) : std::pow(v, weightfactor);
        TRY(h2d(c, c->wm, ss.data(), (size_t)c->M));
        colscale_kernel<<<dim3(blocks), dim3(256), 0, c->stream>>>(c->G, c->ld, c->M, c->wm);
    } else if (c->n_panels > 1) {
        const unsigned blocks = (unsigned)std::min<int64_t>(c->M, (int64_t)c->cus * 16);
        colnorm_kernel<<<dim3(blocks), dim3(256), 0, c->stream>>>(c->G, c->ld, c->M, weightfactor, c->wm);
        colscale_kernel<<<dim3(blocks), dim3(256), 0, c->stream>>>(c->G, c->ld, c->M, c->wm);
    } else {
        weight_fn f = weight_for(c);
        const int threads = (c->TW == 1 ? 4 : c->TW) * 64;
        hipLaunchKernelGGL(f, dim3(c->grid), dim3(threads), 0, c->stream, c->G, c->ld, c->M,
                           c->cols_per_team, c->n_teams_sweep, weightfactor, c->wm);
    }
    HIPCHK(c, hipGetLastError());
    std::vector<double> w((size_t)c->M);
    TRY(d2h(c, w.data(), c->wm, (size_t)c->M));
    if (wm_out) memcpy(wm_out, w.data(), sizeof(double) * (size_t)c->M);
    for (auto &v : w) v = v * v;  // diag(WmSquare) = ADiag * ADiag (potential.py:253)
    TRY(h2d(c, c->wm2, w.data(), (size_t)c->M));
    c->weighted = true;
    c->chain_ready = false;
    c->bt.ready = false;
    return GH_OK;
}

int gh_set_data(gh_ctx *c, const double *dobs, const double *grav_fix)
{
    if (!c || !dobs) return fail(c, GH_ERR_ARG, "gh_set_data: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    const size_t N = (size_t)c->N;
    TRY(dalloc(c, &c->dobs_c, (size_t)c->ld));
    TRY(dalloc(c, &c->gfix, (size_t)c->ld));
    // dobs - mean(dobs) (potential.py:706); numpy's mean is a pairwise sum
    std::vector<double> t(dobs, dobs + N);
    // pairwise summation with numpy's blocking (8-way unrolled blocks of 128)
    struct PW {
        static double sum(const double *a, size_t n)
        {
            if (n < 8) {
                double r = 0.0;
                for (size_t i = 0; i < n; ++i) r += a[i];
                return r;
            }
            if (n <= 128) {
                double r[8];
                for (int k = 0; k < 8; ++k) r[k] = a[k];
                size_t i = 8;
                for (; i + 8 <= n; i += 8)
                    for (int k = 0; k < 8; ++k) r[k] += a[i + k];
                double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < n; ++i) res += a[i];
                return res;
            }
            size_t n2 = n / 2;
            n2 -= n2 % 8;
            return sum(a, n2) + sum(a + n2, n - n2);
        }
    };
    double mean = PW::sum(t.data(), N) / (double)N;
    if (shard_rows(c)) {
        // (row blocks: the mean of ALL observations -- the sum of the ranks' pairwise sums)
        double v2[1] = {PW::sum(t.data(), N)};
        TRY(comm_allreduce_host(c, v2, 1));
        mean = v2[0] / (double)c->sh.N_global;
    }
    for (auto &v : t) v -= mean;
    TRY(h2d(c, c->dobs_c, t.data(), N));
    c->have_fix = grav_fix != nullptr;
    c->gfix_sum = grav_fix ? PW::sum(grav_fix, N) : 0.0;
    if (grav_fix) TRY(h2d(c, c->gfix, grav_fix, N));
    c->have_data = true;
    c->chain_ready = false;
    c->bt.ready = false;
    return GH_OK;
}

int gh_set_reg(gh_ctx *c, int kind, double alpha, double beta, const int shape3[3], const double *mwapr)
{
    if (!c || !mwapr) return fail(c, GH_ERR_ARG, "gh_set_reg: null pointer");
    if (kind < 0 || kind > 3)
        return fail(c, GH_ERR_ARG, "Please choose regularization from 'MS','Damping', 'Smoothness', 'TV'.");
    const bool stencil = (kind == GH_REG_SMOOTHNESS || kind == GH_REG_TV);
    if (shard_cols(c) && stencil) {
        // the finite-difference stencil crosses the shard boundaries: shards of whole z-planes
        if (!shape3 || (int64_t)shape3[0] * shape3[1] * shape3[2] != c->sh.M_global)
            return fail(c, GH_ERR_ARG, "gh_set_reg: Smoothness/TV on a sharded model need the GLOBAL shape nz*ny*nx == M_global");
        const int64_t P = (int64_t)shape3[1] * shape3[2];
        if (c->M < P || c->M % P != 0 || c->sh.m0 % P != 0)
            return fail(c, GH_ERR_UNSUPPORTED, "Smoothness/TV on a sharded model need shards of whole z-planes "
                                               "(%lld cells each): partition the cells with that alignment", (long long)P);
    } else if (stencil) {
        if (!shape3 || (int64_t)shape3[0] * shape3[1] * shape3[2] != c->M)
            return fail(c, GH_ERR_ARG, "gh_set_reg: Smoothness/TV need shape nz*ny*nx == M (carved meshes are not supported by the finite-difference operator)");
    }
    if (kind == GH_REG_MS) TRY(need(c, c->weighted, "gh_set_reg: MS needs gh_weight first (uses Wm^2)"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(dalloc(c, &c->mwapr, (size_t)c->M));
    TRY(dalloc(c, &c->wm2, (size_t)c->M));
    TRY(h2d(c, c->mwapr, mwapr, (size_t)c->M));
    c->reg_kind = kind;
    c->alpha = alpha;
    c->beta = beta;
    if (shape3) {
        c->shape[0] = shape3[0];
        c->shape[1] = shape3[1];
        c->shape[2] = shape3[2];
    }
    c->sh.halo = false;
    if (shard_cols(c) && stencil) {
        gh_ctx::Shard &sh = c->sh;
        const int64_t P = (int64_t)shape3[1] * shape3[2];
        const size_t need = (size_t)c->ld + 8 + 2 * (size_t)sh.world * (size_t)P;
        if (need > sh.buf_n) {
            sh.buf = nullptr;  // (the old block stays in the allocation list until gh_destroy)
            TRY(dalloc(c, &sh.buf, need));
            if (sh.hbuf) hipHostFree(sh.hbuf);
            sh.hbuf = nullptr;
            HIPCHK(c, hipHostMalloc((void **)&sh.hbuf, sizeof(double) * need));
            sh.buf_n = need;
        }
        if (sh.P != P) {
            sh.alo = sh.ahi = nullptr;
            TRY(dalloc(c, &sh.alo, (size_t)P));
            TRY(dalloc(c, &sh.ahi, (size_t)P));
        }
        TRY(dalloc(c, &sh.rb, 2));
        sh.P = P;
        // boundary planes of the prior model, once (collective: every rank is in this call)
        double *hb = sh.buf + c->ld + 8;
        const int64_t nh = 2 * (int64_t)sh.world * P;
        halo_pack_kernel<<<dim3((unsigned)std::min<int64_t>(1024, (nh + 255) / 256)), dim3(256), 0, c->stream>>>(
            c->mwapr, c->M, P, sh.rank, sh.world, hb);
        TRY(comm_allreduce(c, sh.buf, (int64_t)c->ld + 8 + nh));
        if (sh.rank > 0)
            HIPCHK(c, hipMemcpyAsync(sh.alo, hb + ((int64_t)(sh.rank - 1) * 2 + 1) * P, sizeof(double) * (size_t)P,
                                     hipMemcpyDeviceToDevice, c->stream));
        if (sh.rank + 1 < sh.world)
            HIPCHK(c, hipMemcpyAsync(sh.ahi, hb + (int64_t)(sh.rank + 1) * 2 * P, sizeof(double) * (size_t)P,
                                     hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        sh.halo = true;
    }
    c->have_reg = true;
    c->chain_ready = false;
    c->bt.ready = false;
    return GH_OK;
}

int gh_forward(gh_ctx *c, const double *mw, double *dpre)
{
    if (!c || !mw || !dpre) return fail(c, GH_ERR_ARG, "gh_forward: null pointer");
    TRY(need(c, c->have_G, "gh_forward: no kernel matrix resident"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_work(c));
    TRY(h2d(c, c->tmpM, mw, (size_t)c->M));
    SweepArgs a{};
    a.mode = SW_FWD;
    a.x_in = c->tmpM;
    a.slab = c->slab;
    TRY(launch_sweep(c, a));
    reduce_slab(c, nullptr, c->tmpN);
    HIPCHK(c, hipGetLastError());
    if (!shard_rows(c)) TRY(comm_allreduce(c, c->tmpN, c->ld));  // (row blocks: the local rows are complete)
    return d2h(c, dpre, c->tmpN, (size_t)c->N);  // (forward-only sweeps never run on teams)
}

int gh_adjoint(gh_ctx *c, const double *r, double *g)
{
    if (!c || !r || !g) return fail(c, GH_ERR_ARG, "gh_adjoint: null pointer");
    TRY(need(c, c->have_G, "gh_adjoint: no kernel matrix resident"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_work(c));
    HIPCHK(c, hipMemsetAsync(c->tmpN, 0, sizeof(double) * (size_t)c->ld, c->stream));
    TRY(h2d(c, c->tmpN, r, (size_t)c->N));
    SweepArgs a{};
    a.mode = SW_ADJ | SW_GOUT;
    a.r = c->tmpN;
    a.g_out = c->tmpM;
    TRY(launch_sweep(c, a));
    TRY(d2h(c, g, c->tmpM, (size_t)c->M));
    for (int64_t j = 0; j < c->M; ++j) g[j] *= 0.5;  // the sweep writes 2*<G_j, r>
    return GH_OK;
}

int gh_misfit_and_grad(gh_ctx *c, const double *x, double out3[3], double *grad, double *dpre)
{
    if (!c || !x || !out3 || !grad) return fail(c, GH_ERR_ARG, "gh_misfit_and_grad: null pointer");
    TRY(need(c, c->have_G && c->have_data && c->have_reg,
             "gh_misfit_and_grad: needs a kernel matrix, gh_set_data and gh_set_reg"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_work(c));
    const gh_ctx::StateSet &o = c->st[3];
    TRY(h2d(c, c->xb[3], x, (size_t)c->M));
    TRY(eval_forward(c, c->xb[3], o));
    SweepArgs a{};
    a.mode = SW_ADJ | SW_GOUT;
    a.r = o.r;
    a.greg = o.greg;
    a.g_out = c->tmpM;
    TRY(launch_sweep(c, a));
    TRY(scal_ready(c, o));
    HIPCHK(c, hipMemcpyAsync(c->h_scal, o.scal, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRY(d2h(c, grad, c->tmpM, (size_t)c->M));
    if (dpre) TRY(d2h(c, dpre, o.d, (size_t)c->N));
    TRY(lonsym_epilogue_check(c));
    out3[0] = c->h_scal[2];
    out3[1] = c->h_scal[0];
    out3[2] = c->h_scal[1];
    return GH_OK;
}

int gh_reg_eval(gh_ctx *c, int kind, double beta, const int shape3[3], int ms_grad_den_mw, const double *mw,
                const double *mwapr, double *value, double *grad)
{
    if (!c || !mw || !mwapr || !value) return fail(c, GH_ERR_ARG, "gh_reg_eval: null pointer");
    if (kind < 0 || kind > 3)
        return fail(c, GH_ERR_ARG, "Please choose regularization from 'MS','Damping', 'Smoothness', 'TV'.");
    if (kind == GH_REG_SMOOTHNESS || kind == GH_REG_TV)
        if (!shape3 || (int64_t)shape3[0] * shape3[1] * shape3[2] != c->M)
            return fail(c, GH_ERR_ARG, "gh_reg_eval: Smoothness/TV need shape nz*ny*nx == M");
    if (kind == GH_REG_MS) TRY(need(c, c->weighted, "gh_reg_eval: MS needs gh_weight first (uses Wm^2)"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_work(c));
    double *dx = c->xb[3], *dapr = c->st[3].greg, *dg = c->tmpM;
    TRY(h2d(c, dx, mw, (size_t)c->M));
    TRY(h2d(c, dapr, mwapr, (size_t)c->M));
    RegArgs ra{};
    ra.ms_grad_den_mw = ms_grad_den_mw;
    ra.kind = kind;
    ra.M = c->M;
    ra.nz = shape3 ? shape3[0] : 1;
    ra.ny = shape3 ? shape3[1] : 1;
    ra.nx = shape3 ? shape3[2] : (int)c->M;
    ra.alpha = 1.0;
    ra.beta = beta;
    ra.x = dx;
    ra.mwapr = dapr;
    ra.wm2 = c->wm2;
    ra.greg = dg;
    ra.regpart = c->regpart;
    reg_kernel<<<dim3(c->n_regpart), dim3(256), 0, c->stream>>>(ra);
    sum_kernel<<<dim3(1), dim3(1024), 0, c->stream>>>(c->regpart, c->n_regpart, c->st[3].scal);
    HIPCHK(c, hipGetLastError());
    TRY(d2h(c, c->h_scal, c->st[3].scal, 2));
    *value = c->h_scal[0];
    if (grad) TRY(d2h(c, grad, dg, (size_t)c->M));
    return GH_OK;
}

int gh_compress_wavelet(gh_ctx *c, int dims, const int shape3[3], double thr, int levels,
                        int64_t *nnz_out, int64_t *ncols_out)
{
    if (!c) return GH_ERR_ARG;
    TRY(need(c, c->have_G && c->weighted, "gh_compress_wavelet: needs the weighted kernel (gh_weight) first"));
    if (dims != 1 && dims != 3) return fail(c, GH_ERR_ARG, "gh_compress_wavelet: dims must be 1 or 3");
    if (levels < 1 || levels > 4) return fail(c, GH_ERR_ARG, "gh_compress_wavelet: levels must be 1..4");
    if (!(thr >= 0)) return fail(c, GH_ERR_ARG, "gh_compress_wavelet: threshold must be >= 0");
    gh_ctx::Wavelet &w = c->wv;
    if (w.on || w.indptr) return fail(c, GH_ERR_ARG, "gh_compress_wavelet: already compressed");
    // (row blocks hold whole rows of the kernel: the compressor works row by row, compressor3D.py:17-44, so every rank
    // compresses its own rows; column blocks would have to exchange the rows first)
    if (c->sh.kind != 0 && !shard_rows(c))
        return fail(c, GH_ERR_UNSUPPORTED, "wavelet forward on a kernel sharded in column blocks is not supported (row blocks: gh_shard_init_rows)");
    if (dims == 3) {
        if (!shape3 || (int64_t)shape3[0] * shape3[1] * shape3[2] != c->M)
            return fail(c, GH_ERR_ARG, "cannot reshape array of size %lld into shape (%d,%d,%d)",
                        (long long)c->M, shape3 ? shape3[0] : 0, shape3 ? shape3[1] : 0, shape3 ? shape3[2] : 0);
        for (int k = 0; k < 3; ++k) {
            w.shape[k] = shape3[k];
            w.tax[k] = true;
        }
    } else {
        if (c->M > 0x7fffffffLL) return fail(c, GH_ERR_UNSUPPORTED, "model too long");
        w.shape[0] = w.shape[1] = 1;
        w.shape[2] = (int)c->M;
        w.tax[0] = w.tax[1] = false;
        w.tax[2] = true;
    }
    w.dims = dims;
    w.levels = levels;
    w.thr = thr;
    wavelet_plan(w);
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_work(c));
    const int64_t N = c->N, M = c->M, Mp = w.Mp;
    // row chunks: 4 buffers of chunk x Mp doubles, bounded to ~2 GB in total
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(N, (int64_t)(512 << 20) / (Mp * 8)));
    double *X = nullptr, *C = nullptr, *S1 = nullptr, *S2 = nullptr;
    int *count = nullptr;
    HIPCHK(c, hipMalloc((void **)&X, sizeof(double) * (size_t)(chunk * M)));
    HIPCHK(c, hipMalloc((void **)&C, sizeof(double) * (size_t)(chunk * Mp)));
    HIPCHK(c, hipMalloc((void **)&S1, sizeof(double) * (size_t)(chunk * Mp)));
    HIPCHK(c, hipMalloc((void **)&S2, sizeof(double) * (size_t)(chunk * Mp)));
    HIPCHK(c, hipMalloc((void **)&count, sizeof(int) * (size_t)N));
    TRY(dalloc(c, &w.indptr, (size_t)N + 1));
    std::vector<int> hcount((size_t)N);
    std::vector<int64_t> hptr((size_t)N + 1, 0);
    int rc = GH_OK;
    for (int pass = 0; pass < 2 && rc == GH_OK; ++pass) {
        if (pass == 1) {
            HIPCHK(c, hipMemcpyAsync(hcount.data(), count, sizeof(int) * (size_t)N, hipMemcpyDeviceToHost,
                                     c->stream));
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int64_t i = 0; i < N; ++i) hptr[i + 1] = hptr[i] + hcount[i];
            w.nnz = hptr[N];
            if (w.nnz > 0x7fffffff00LL) return fail(c, GH_ERR_UNSUPPORTED, "too many non-zeros");
            TRY(dalloc(c, &w.indices, (size_t)std::max<int64_t>(w.nnz, 1), false));
            TRY(dalloc(c, &w.data, (size_t)std::max<int64_t>(w.nnz, 1), false));
            HIPCHK(c, hipMemcpyAsync(w.indptr, hptr.data(), sizeof(int64_t) * (size_t)(N + 1),
                                     hipMemcpyHostToDevice, c->stream));
        }
        for (int64_t i0 = 0; i0 < N; i0 += chunk) {
            const int64_t nr = std::min(chunk, N - i0);
            if (c->mf) {
                // no stored kernel: the rows are evaluated (twice over the two passes: setup path)
                for (int64_t r0 = 0; r0 < nr; r0 += 32768) {
                    const int64_t rn = std::min<int64_t>(32768, nr - r0);
                    mf_rows_kernel<<<dim3((unsigned)((M + 255) / 256), (unsigned)rn), dim3(256), 0, c->stream>>>(
                        mf_geom(c), c->wm, i0 + r0, rn, X + r0 * M);
                }
            } else {
                gather_rows_kernel<<<dim3((unsigned)((M + 31) / 32), (unsigned)((nr + 31) / 32)), dim3(256), 0,
                                     c->stream>>>(c->G, c->ld, M, i0, nr, X);
            }
            HIPCHK(c, hipMemsetAsync(C, 0, sizeof(double) * (size_t)(nr * Mp), c->stream));
            rc = run_dwt(c, X, M, nr, C, S1, S2);
            if (rc != GH_OK) break;
            if (pass == 0)
                csr_count_kernel<<<dim3((unsigned)nr), dim3(256), 0, c->stream>>>(C, Mp, thr, count + i0);
            else
                csr_fill_kernel<<<dim3((unsigned)nr), dim3(256), 0, c->stream>>>(C, Mp, thr, w.indptr, i0,
                                                                                 w.indices, w.data);
        }
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(X);
    hipFree(C);
    hipFree(S1);
    hipFree(S2);
    hipFree(count);
    if (rc != GH_OK) return rc;
    if (e != hipSuccess) return fail(c, GH_ERR_HIP, "gh_compress_wavelet: %s", hipGetErrorString(e));
    // (a second compression of the same context -- other dims / levels / shape, other Mp -- gets fresh,
    // zeroed scratch: the one-launch transform never writes the gaps of the packed layout and relies
    // on their being zero; the old blocks stay in the allocation list until gh_destroy)
    if (w.coeff_n != Mp) w.coeff = w.s1 = w.s2 = nullptr;
    TRY(dalloc(c, &w.coeff, (size_t)Mp));
    TRY(dalloc(c, &w.s1, (size_t)Mp));
    TRY(dalloc(c, &w.s2, (size_t)Mp));
    w.coeff_n = Mp;
    HIPCHK(c, hipMemsetAsync(w.coeff, 0, sizeof(double) * (size_t)Mp, c->stream));
    wavelet_plan_lds(c);
    w.on = true;
    w.F_valid = false;
    if (!shard_rows(c)) c->rs.state = 0;  // plan the resident chain kernel again (it would need the dense form)
    if (c->ls) c->ls->res.state = 0;  // (the persistent harmonic pass has no compressed forward: planned again, refused)
    c->chain_ready = false;
    c->bt.ready = false;
    if (nnz_out) *nnz_out = w.nnz;
    if (ncols_out) *ncols_out = Mp;
    return GH_OK;
}

int gh_download_csr(gh_ctx *c, int64_t *indptr, int32_t *indices, double *data)
{
    if (!c || !indptr || !indices || !data) return fail(c, GH_ERR_ARG, "gh_download_csr: null pointer");
    TRY(need(c, c->wv.on, "gh_download_csr: call gh_compress_wavelet first"));
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpyAsync(indptr, c->wv.indptr, sizeof(int64_t) * (size_t)(c->N + 1), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(indices, c->wv.indices, sizeof(int) * (size_t)c->wv.nnz, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(data, c->wv.data, sizeof(double) * (size_t)c->wv.nnz, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

int gh_model_coeffs(gh_ctx *c, const double *mw, double *coeff)
{
    if (!c || !mw || !coeff) return fail(c, GH_ERR_ARG, "gh_model_coeffs: null pointer");
    TRY(need(c, c->wv.on, "gh_model_coeffs: call gh_compress_wavelet first"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(h2d(c, c->tmpM, mw, (size_t)c->M));
    HIPCHK(c, hipMemsetAsync(c->wv.coeff, 0, sizeof(double) * (size_t)c->wv.Mp, c->stream));
    TRY(run_dwt(c, c->tmpM, c->M, 1, c->wv.coeff, c->wv.s1, c->wv.s2));
    return d2h(c, coeff, c->wv.coeff, (size_t)c->wv.Mp);
}

int gh_forward_wavelet(gh_ctx *c, const double *mw, double *dpre)
{
    if (!c || !mw || !dpre) return fail(c, GH_ERR_ARG, "gh_forward_wavelet: null pointer");
    TRY(need(c, c->wv.on, "gh_forward_wavelet: call gh_compress_wavelet first"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(h2d(c, c->tmpM, mw, (size_t)c->M));
    TRY(wavelet_forward(c, c->tmpM, c->tmpN));
    return d2h(c, dpre, c->tmpN, (size_t)c->N);
}

int gh_chain_init(gh_ctx *c, const double *x0, const double *low, const double *high)
{
    if (!c || !x0 || !low || !high) return fail(c, GH_ERR_ARG, "gh_chain_init: null pointer");
    TRY(need(c, c->have_G && c->have_data && c->have_reg,
             "gh_chain_init: needs a kernel matrix, gh_set_data and gh_set_reg"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_work(c));
    c->cur = 0;
    c->xcur = 0;
    c->spec_valid = c->pn_valid = false;
    c->st_stale = false;
    c->accept_count = 0;
    TRY(h2d(c, c->xb[0], x0, (size_t)c->M));
    TRY(h2d(c, c->low, low, (size_t)c->M));
    TRY(h2d(c, c->high, high, (size_t)c->M));
    TRY(eval_forward(c, c->xb[0], c->st[0]));
    TRY(scal_ready(c, c->st[0]));
    TRY(d2h(c, c->h_scal, c->st[0].scal, 4));
    TRY(lonsym_epilogue_check(c));
    c->U_cur[0] = c->h_scal[2];
    c->U_cur[1] = c->h_scal[0];
    c->U_cur[2] = c->h_scal[1];
    c->chain_ready = true;
    return GH_OK;
}

int gh_chain_prefetch_momentum(gh_ctx *c, const double *p0_next)
{
    if (!c || !p0_next) return fail(c, GH_ERR_ARG, "gh_chain_prefetch_momentum: null pointer");
    TRY(need(c, c->chain_ready, "gh_chain_prefetch_momentum: call gh_chain_init first"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(h2d(c, c->pn, p0_next, (size_t)c->M));
    // initial kinetic energy of that trajectory, summed exactly like the non-speculative path
    sumsq_kernel<<<dim3(c->n_pp0), dim3(256), 0, c->stream>>>(c->pn, c->M, c->pp0_part);
    {
        std::vector<double> part((size_t)c->n_pp0);
        TRY(d2h(c, part.data(), c->pp0_part, (size_t)c->n_pp0));
        double s = 0.0;
        for (double v : part) s += v;
        c->pn_pp0 = s;
    }
    c->pn_probe[0] = p0_next[0];
    c->pn_probe[1] = p0_next[c->M / 2];
    c->pn_probe[2] = p0_next[c->M - 1];
    c->pn_valid = true;
    return GH_OK;
}

// d, r and the scalars of the current sample after launches of the resident chain kernel
static int chain_state_fresh(gh_ctx *c)
{
    if (!c->st_stale) return GH_OK;
    TRY(eval_forward(c, c->xb[c->xcur], c->st[c->cur]));
    TRY(scal_ready(c, c->st[c->cur]));
    TRY(d2h(c, c->h_scal, c->st[c->cur].scal, 4));
    c->U_cur[0] = c->h_scal[2];
    c->U_cur[1] = c->h_scal[0];
    c->U_cur[2] = c->h_scal[1];
    c->st_stale = false;
    return GH_OK;
}

static inline int other_of3(int a, int b)
{
    for (int i = 0; i < 3; ++i)
        if (i != a && i != b) return i;
    return 0;
}

// p0_next (or nullptr): momentum of the NEXT trajectory, valid until this call returns.  Announced
// this way (gh_chain_run) it is uploaded on a second stream while this trajectory's sweeps run --
// through the public gh_chain_prefetch_momentum the upload happens before them, with the GPU idle.
static int chain_trajectory_impl(gh_ctx *c, const double *p0, double dt, int L, double u, const double *p0_next,
                                 int *accepted, double out5[5])
{
    if (!c || !p0 || !accepted || !out5) return fail(c, GH_ERR_ARG, "gh_chain_trajectory: null pointer");
    TRY(need(c, c->chain_ready, "gh_chain_trajectory: call gh_chain_init first"));
    if (L < 1) return fail(c, GH_ERR_ARG, "gh_chain_trajectory: L must be >= 1");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(chain_state_fresh(c));
    // (sweeps launched from here may run on teams of workgroups whose time-out this function handles)
    struct TeamsOk {
        gh_ctx *c;
        ~TeamsOk() { c->chain_teams_ok = false; }
    } teams_guard{c};
    c->chain_teams_ok = true;
    const size_t M = (size_t)c->M;
    const int nt = c->n_teams;
    // Was the first step of this trajectory already taken speculatively by the previous call's
    // last sweep (same momentum, same dt, previous proposal accepted)?
    const bool use_spec = c->spec_valid && c->spec_dt == dt && p0[0] == c->spec_probe[0] &&
                          p0[c->M / 2] == c->spec_probe[1] && p0[c->M - 1] == c->spec_probe[2];
    int xin, pin, sin, s0;
    if (use_spec) {
        xin = c->spec_x;
        pin = c->spec_p;
        sin = c->spec_set;
        s0 = 1;
        c->spec_hits += 1;
    } else {
        // momentum upload + kinetic energy of p0 (hmc.py:95-104)
        HIPCHK(c, hipMemcpyAsync(c->pb[0], p0, M * sizeof(double), hipMemcpyHostToDevice, c->stream));
        sumsq_kernel<<<dim3(c->n_pp0), dim3(256), 0, c->stream>>>(c->pb[0], c->M, c->pp0_part);
        xin = c->xcur;
        pin = 0;
        sin = c->cur;
        s0 = 0;
        if (c->spec_valid) c->spec_misses += 1;
    }
    c->spec_valid = false;
    for (int s = s0; s < L; ++s) {
        const int xout = other_of3(c->xcur, xin);
        const int sout = (sin != c->cur) ? sin : other_of3(c->cur, c->cur);
        SweepArgs a{};
        a.mode = SW_ADJ | SW_UPD | (c->wv.on ? 0 : SW_FWD);
        a.r = c->st[sin].r;
        a.greg = c->st[sin].greg;
        a.x_in = c->xb[xin];
        a.p_in = c->pb[pin];
        a.x_out = c->xb[xout];
        a.p_out = c->pb[pin ^ 1];
        a.low = c->low;
        a.high = c->high;
        a.c_u = (s == 0) ? dt * 0.5 : dt;
        a.dt = dt;
        a.slab = c->slab;
        TRY(launch_sweep(c, a));
        TRY(finalize(c, c->xb[xout], c->st[sout]));
        xin = xout;
        pin ^= 1;
        sin = sout;
    }
    // Last half step of the momentum + kinetic energy (hmc.py:151-157).  When the caller has
    // announced the next trajectory's momentum, the same sweep also takes that trajectory's
    // first leapfrog step from the proposal (valid if the proposal is accepted): the gradient
    // at the proposal is needed by both, so the extra sweep per trajectory disappears.
    bool spec = c->pn_valid;
    double probe[3] = {c->pn_probe[0], c->pn_probe[1], c->pn_probe[2]};
    double pn_pp0 = c->pn_pp0;
    bool pn_deferred = false;
    if (p0_next) {
        if (!c->copy_stream) {
            HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
            HIPCHK(c, hipEventCreateWithFlags(&c->copy_ev, hipEventDisableTiming));
        }
        // (c->pn was last read by the previous trajectory's final sweep, which has completed)
        HIPCHK(c, hipMemcpyAsync(c->pn, p0_next, M * sizeof(double), hipMemcpyHostToDevice, c->copy_stream));
        HIPCHK(c, hipEventRecord(c->copy_ev, c->copy_stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->copy_ev, 0));
        // its initial kinetic energy, summed exactly like gh_chain_prefetch_momentum does
        sumsq_kernel<<<dim3(c->n_pp0), dim3(256), 0, c->stream>>>(c->pn, c->M, c->pn0_part);
        probe[0] = p0_next[0];
        probe[1] = p0_next[c->M / 2];
        probe[2] = p0_next[c->M - 1];
        spec = true;
        pn_deferred = true;
    }
    const int xs = other_of3(c->xcur, xin), ss = other_of3(c->cur, sin);
    {
        SweepArgs a{};
        a.mode = SW_ADJ | SW_PFIN;
        a.r = c->st[sin].r;
        a.greg = c->st[sin].greg;
        a.p_in = c->pb[pin];
        a.p_out = c->pb[pin ^ 1];
        a.c_p = dt * 0.5;
        a.pp_part = c->pp_part;
        if (spec) {
            a.mode |= SW_SPEC | SW_UPD | (c->wv.on ? 0 : SW_FWD);
            a.pn_in = c->pn;
            a.x_in = c->xb[xin];
            a.x_out = c->xb[xs];
            a.low = c->low;
            a.high = c->high;
            a.c_u = dt * 0.5;
            a.dt = dt;
            a.slab = c->slab;
        }
        TRY(launch_sweep(c, a));
        if (spec) TRY(finalize(c, c->xb[xs], c->st[ss]));
    }
    double *h = c->h_scal;
    TRY(scal_ready(c, c->st[sin]));
    if (spec) TRY(scal_ready(c, c->st[ss]));
    HIPCHK(c, hipMemcpyAsync(h, c->st[sin].scal, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h + 16, c->pp_part, (size_t)nt * sizeof(double), hipMemcpyDeviceToHost,
                             c->stream));
    if (spec) {
        HIPCHK(c, hipMemcpyAsync(h + 4, c->st[ss].scal, 4 * sizeof(double), hipMemcpyDeviceToHost,
                                 c->stream));
    }
    if (!use_spec)
        HIPCHK(c, hipMemcpyAsync(h + 16 + 2 * nt, c->pp0_part, (size_t)c->n_pp0 * sizeof(double),
                                 hipMemcpyDeviceToHost, c->stream));
    if (pn_deferred)
        HIPCHK(c, hipMemcpyAsync(h + 16 + 2 * nt + c->n_pp0, c->pn0_part, (size_t)c->n_pp0 * sizeof(double),
                                 hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // a team sweep of this trajectory gave up (its workgroups were not all resident): nothing of
    // the chain's current state was touched -- run the trajectory again, in row panels.  Sharded
    // chain: decided by all ranks together, below (the flag rides on the scalar all-reduce)
    TRY(lonsym_epilogue_check(c));
    bool failed = false;
    TRY(team_failed(c, &failed));
    if (!failed) TRY(mft_failed(c, &failed));  // (matrix-free chain on teams: same contract)
    auto redo = [&]() -> int {
        c->spec_valid = c->pn_valid = false;
        const int rc = chain_trajectory_impl(c, p0, dt, L, u, p0_next, accepted, out5);
        team_resume(c);
        return rc;
    };
    if (failed && c->sh.kind == 0) return redo();
    if (pn_deferred) {
        double s = 0.0;
        for (int t = 0; t < c->n_pp0; ++t) s += h[16 + 2 * nt + c->n_pp0 + t];
        pn_pp0 = s;
    }
    double pp1 = 0.0, pp0 = 0.0;
    for (int t = 0; t < nt; ++t) pp1 += h[16 + t];
    if (use_spec)
        pp0 = c->spec_pp0;
    else
        for (int t = 0; t < c->n_pp0; ++t) pp0 += h[16 + 2 * nt + t];
    double pn_pp0_g = pn_pp0;
    if (shard_cols(c)) {
        // kinetic energies are sums over cells: combine the ranks' parts (same bits everywhere)
        double v[4] = {pp1, use_spec ? 0.0 : pp0, spec ? pn_pp0 : 0.0, failed ? 1.0 : 0.0};
        if (failed) v[0] = v[1] = v[2] = 0.0;  // (whatever the aborted sweeps left: not worth a NaN in the sum)
        TRY(comm_allreduce_host(c, v, 4));
        if (v[3] != 0.0) {
            // some rank's team sweep gave up: its slab went into everybody's d and r through the
            // all-reduces of this trajectory -- every rank repeats it, in row panels, together
            if (!failed) TRY(team_mark_failed(c, "another rank's"));
            return redo();
        }
        pp1 = v[0];
        if (!use_spec) pp0 = v[1];
        pn_pp0_g = v[2];
    }
    const double Unew[3] = {h[2], h[0], h[1]};
    const double Hcur = 0.5 * pp0 + c->U_cur[0];
    const double Hnew = 0.5 * pp1 + Unew[0];
    const bool acc = (Hnew < Hcur) || (u < std::exp(-(Hnew - Hcur)));
    if (acc) {
        c->xcur = xin;
        c->cur = sin;
        c->U_cur[0] = Unew[0];
        c->U_cur[1] = Unew[1];
        c->U_cur[2] = Unew[2];
        if (spec) {
            c->spec_valid = true;
            c->spec_dt = dt;
            c->spec_pp0 = pn_pp0_g;
            c->spec_x = xs;
            c->spec_p = pin ^ 1;
            c->spec_set = ss;
            c->spec_probe[0] = probe[0];
            c->spec_probe[1] = probe[1];
            c->spec_probe[2] = probe[2];
        }
    } else if (spec) {
        c->spec_misses += 1;  // the speculative step belonged to a rejected proposal
    }
    c->pn_valid = false;
    *accepted = acc ? 1 : 0;
    out5[0] = c->U_cur[0];
    out5[1] = c->U_cur[1];
    out5[2] = c->U_cur[2];
    out5[3] = Hcur;
    out5[4] = Hnew;
    return GH_OK;
}

int gh_chain_trajectory(gh_ctx *c, const double *p0, double dt, int L, double u, int *accepted,
                        double out5[5])
{
    return chain_trajectory_impl(c, p0, dt, L, u, nullptr, accepted, out5);
}

int gh_chain_run(gh_ctx *c, int K, const int *L, const double *p0s, const double *us, double dt,
                 const double *p0_lookahead, int64_t stop_at_accepts, int64_t record_from, int *accepted,
                 double *out5s, double *x_out, int *n_run)
{
    if (!c || K < 1 || !L || !p0s || !us || !accepted || !out5s || !n_run)
        return fail(c, GH_ERR_ARG, "gh_chain_run: bad arguments");
    TRY(need(c, c->chain_ready, "gh_chain_run: call gh_chain_init first"));
    const size_t M = (size_t)c->M;
    *n_run = 0;
    // already there (a caller that submits batches ahead of looking at the results)
    if (stop_at_accepts > 0 && c->accept_count >= stop_at_accepts) return GH_OK;
    if (lonres_usable(c)) {
        // the shift-invariant store in the harmonic domain: the whole batch in one persistent launch (lonres.hip.h)
        int64_t steps = 0;
        for (int k = 0; k < K; ++k) {
            if (L[k] < 1) return fail(c, GH_ERR_ARG, "gh_chain_run: L must be >= 1");
            steps += L[k];
        }
        if (steps < ((int64_t)1 << 28)) {
            const int rc = chain_run_lonres(c, c, K, L, p0s, nullptr, us, dt, stop_at_accepts, record_from, accepted, out5s, x_out, n_run);
            if (rc != GH_RESIDENT_ABORTED) return rc;
            *n_run = 0;
        }
    }
    if (resident_usable(c)) {
        int64_t steps = 0;
        bool ok = true;
        for (int k = 0; k < K; ++k) {
            if (L[k] < 1) return fail(c, GH_ERR_ARG, "gh_chain_run: L must be >= 1");
            steps += L[k];
        }
        ok = steps < ((int64_t)1 << 28);  // granule tags are 32-bit
        if (ok) {
            const int rc = chain_run_resident(c, K, L, p0s, us, dt, stop_at_accepts, record_from, accepted,
                                              out5s, x_out, n_run);
            if (rc != GH_RESIDENT_ABORTED) return rc;
            *n_run = 0;
        }
    }
    for (int k = 0; k < K; ++k) {
        const double *nxt = (k + 1 < K) ? p0s + (size_t)(k + 1) * M : p0_lookahead;
        TRY(chain_trajectory_impl(c, p0s + (size_t)k * M, dt, L[k], us[k], nxt, &accepted[k], out5s + 5 * k));
        *n_run = k + 1;
        if (accepted[k]) {
            c->accept_count += 1;
            if (c->ring && c->accept_count > record_from) TRY(gh_posterior_add(c));
            if (x_out)
                HIPCHK(c, hipMemcpyAsync(x_out + (size_t)k * M, c->xb[c->xcur], M * sizeof(double),
                                         hipMemcpyDeviceToHost, c->stream));
            if (stop_at_accepts > 0 && c->accept_count >= stop_at_accepts) break;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

int gh_chain_get_x(gh_ctx *c, double *x)
{
    if (!c || !x) return fail(c, GH_ERR_ARG, "gh_chain_get_x: null pointer");
    TRY(need(c, c->chain_ready, "gh_chain_get_x: call gh_chain_init first"));
    HIPCHK(c, hipSetDevice(c->device));
    return d2h(c, x, c->xb[c->xcur], (size_t)c->M);
}

int gh_chain_get_dsyn(gh_ctx *c, double *dsyn)
{
    if (!c || !dsyn) return fail(c, GH_ERR_ARG, "gh_chain_get_dsyn: null pointer");
    TRY(need(c, c->chain_ready, "gh_chain_get_dsyn: call gh_chain_init first"));
    HIPCHK(c, hipSetDevice(c->device));
    TRY(chain_state_fresh(c));
    return d2h(c, dsyn, c->st[c->cur].d, (size_t)c->N);
}

int gh_chain_stats(gh_ctx *c, int64_t *spec_hits, int64_t *spec_misses)
{
    if (!c) return GH_ERR_ARG;
    if (spec_hits) *spec_hits = c->spec_hits;
    if (spec_misses) *spec_misses = c->spec_misses;
    return GH_OK;
}

int gh_team_sweep_stats(gh_ctx *c, int *members, int64_t *launches, int *timeouts, int64_t *late_parts)
{
    if (!c) return GH_ERR_ARG;
    if (late_parts) *late_parts = c->tm.late_polls;
    if (members) *members = c->tm.state != 0 ? c->tm.Q : 0;
    if (launches) *launches = c->tm.launches;
    if (timeouts) *timeouts = c->tm.aborts;
    return GH_OK;
}

int gh_chain_resident_stats(gh_ctx *c, int64_t *launches, int64_t *evaluations)
{
    if (!c) return GH_ERR_ARG;
    if (launches) *launches = c->rs.launches;
    if (evaluations) *evaluations = c->rs.evals;
    return GH_OK;
}

int gh_posterior_window(gh_ctx *c, int K)
{
    if (!c || K < 1) return fail(c, GH_ERR_ARG, "gh_posterior_window: K must be >= 1");
    if (c->ring) return fail(c, GH_ERR_ARG, "gh_posterior_window: window already allocated");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(dalloc(c, &c->ring, (size_t)K * (size_t)c->M));
    TRY(dalloc(c, &c->ring_mean, (size_t)c->M));
    TRY(dalloc(c, &c->ring_sd, (size_t)c->M));
    c->ring_K = K;
    c->ring_next = 0;
    c->ring_count = 0;
    return GH_OK;
}

int gh_posterior_add(gh_ctx *c)
{
    if (!c) return GH_ERR_ARG;
    TRY(need(c, c->chain_ready && c->ring, "gh_posterior_add: needs gh_chain_init and gh_posterior_window"));
    HIPCHK(c, hipSetDevice(c->device));
    ring_store_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(
        c->xb[c->xcur], c->weighted ? c->wm : nullptr, c->M, c->ring + (size_t)c->ring_next * (size_t)c->M);
    HIPCHK(c, hipGetLastError());
    c->ring_next = (c->ring_next + 1) % c->ring_K;
    c->ring_count += 1;
    return GH_OK;
}

int gh_posterior_read(gh_ctx *c, int64_t *n_in_window, int64_t *n_total, double *mean, double *sd)
{
    if (!c) return GH_ERR_ARG;
    TRY(need(c, c->ring != nullptr, "gh_posterior_read: call gh_posterior_window first"));
    const int nvalid = (int)std::min<int64_t>(c->ring_count, c->ring_K);
    if (n_in_window) *n_in_window = nvalid;
    if (n_total) *n_total = c->ring_count;
    if (nvalid == 0 || (!mean && !sd)) return GH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    ring_stats_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(c->ring, c->M, nvalid,
                                                                                        c->ring_mean, c->ring_sd);
    HIPCHK(c, hipGetLastError());
    if (mean) TRY(d2h(c, mean, c->ring_mean, (size_t)c->M));
    if (sd) TRY(d2h(c, sd, c->ring_sd, (size_t)c->M));
    return GH_OK;
}

// ---- a batch of chains on the shift-invariant store (BASELINE configs[3]: "8 chains"; the reference runs them
// as MPI ranks, example/global/run_main.sh:16, each with its own 4.25 GB kernel).  A step of ONE chain on the
// table keeps a fraction of the chip busy for ~60 us, most of it latency: the chains are C light contexts
// that share the parent's tables (T, T^: read-only) and problem vectors, each with its own stream, chain
// state and work buffers, driven by one host thread each -- their passes overlap on the GPU.
static int kids_make(gh_ctx *c, int C, const double *x0s, const double *low, const double *high)
{
    for (gh_ctx *k : c->kids) gh_destroy(k);
    c->kids.clear();
    for (int i = 0; i < C; ++i) {
        gh_ctx *k = nullptr;
        const int rc0 = gh_create(&k, c->device, c->N, c->M);
        if (rc0 != GH_OK) return fail(c, rc0, "gh_batch_init: %s", gh_last_error(nullptr));
        c->kids.push_back(k);
        k->cell_kind = c->cell_kind;
        k->ratio = c->ratio;
        k->have_obs = k->have_cells = k->have_G = true;
        k->mf = true;
        k->weighted = c->weighted;
        k->wm = c->wm;
        k->wm2 = c->wm2;
        k->dobs_c = c->dobs_c;
        k->gfix = c->gfix;
        k->gfix_sum = c->gfix_sum;
        k->mwapr = c->mwapr;
        k->have_data = c->have_data;
        k->have_fix = c->have_fix;
        k->have_reg = c->have_reg;
        k->reg_kind = c->reg_kind;
        for (int q = 0; q < 3; ++q) k->shape[q] = c->shape[q];
        k->alpha = c->alpha;
        k->beta = c->beta;
        k->ls = new LonSymHost(*c->ls);  // (the tables are the parent's; the pass's work buffers are its own)
        k->ls->Rhat = k->ls->Dpart = nullptr;
        k->ls->res = LonSymHost::Res();
        k->ls->res.state = -1;  // (the persistent pass takes every CU: not for chains that share the GPU)
        k->ls->dbg = nullptr;
        k->ls->csum = nullptr;
        k->ls->epi_abort = nullptr;
        k->ls->rhat_of = nullptr;
        k->ls->post_pending = false;
        k->ls->epi_tag = 0;
        if (k->ls->harm) {
            int rc = dalloc(k, &k->ls->Rhat, (size_t)k->ls->na * (size_t)k->ls->nf);
            if (rc == GH_OK) rc = dalloc(k, &k->ls->Dpart, (size_t)k->ls->hgrid * (size_t)k->ls->na * (size_t)k->ls->nf);
            if (rc == GH_OK && k->ls->fused) rc = dalloc(k, &k->ls->csum, 2 * (size_t)k->ls->na);
            if (rc == GH_OK && k->ls->fused) rc = dalloc(k, &k->ls->epi_abort, 4);
            if (rc != GH_OK) return fail(c, rc, "gh_batch_init: %s", gh_last_error(k));
        }
        if (k->ls->wide) {
            k->ls->Xhat = nullptr;
            int rc = dalloc(k, &k->ls->Rhat, (size_t)k->ls->na * (size_t)k->ls->nfp);
            if (rc == GH_OK) rc = dalloc(k, &k->ls->Xhat, (size_t)k->ls->nc * (size_t)k->ls->nf);
            if (rc == GH_OK) rc = dalloc(k, &k->ls->Dpart, (size_t)k->ls->wparts * (k->ls->wmirror ? 2 : 1) * (size_t)k->ls->na * (size_t)k->ls->nfp);
            if (rc != GH_OK) return fail(c, rc, "gh_batch_init: %s", gh_last_error(k));
        }
        int rc = configure_mf(k);
        if (rc == GH_OK) rc = dalloc(k, &k->mf_stats, 1);
        if (rc == GH_OK) rc = gh_chain_init(k, x0s + (size_t)i * (size_t)c->M, low, high);
        if (rc != GH_OK) return fail(c, rc, "gh_batch_init (chain %d): %s", i, gh_last_error(k));
    }
    c->bt.C = C;
    c->bt.ready = true;
    c->bt.run = gh_ctx::Batch::Run();
    return GH_OK;
}

// T trajectories of every chain (lists chain-major), each chain on its own thread and stream; results of
// chain i in slots i * Tout + t.  Nothing stays in flight.
static int kids_run(gh_ctx *c, int T, const int *L, const double *const *p0rows, const double *p0flat, const double *us,
                    double dt, int *accepted, double *out5s, double *x_out, int Tout)
{
    const int C = (int)c->kids.size();
    const size_t M = (size_t)c->M;
    std::vector<int> rcs((size_t)C, GH_OK);
    // On the harmonic store the chains TAKE TURNS in the persistent launch (lonres.hip.h) -- a
    // chain's whole list in one launch with the table in the workgroups' registers: one chain alone runs faster that way
    // (35 k steps/s at C4) than eight side by side on the launches per phase (29-32 k together).  A launch that gives up
    // leaves its chain untouched: that chain and the ones behind it run on their own streams as below.
    int first_threaded = 0;
    if (T > 0 && lonres_usable(c)) {
        for (int i = 0; i < C; ++i) {
            gh_ctx *k = c->kids[(size_t)i];
            int n_run = 0;
            const size_t o = (size_t)i * (size_t)Tout;
            const int rc = chain_run_lonres(c, k, T, L + (size_t)i * T, p0flat ? p0flat + (size_t)i * T * M : nullptr,
                                            p0rows ? p0rows + (size_t)i * T : nullptr, us + (size_t)i * T, dt, 0, 0, accepted + o,
                                            out5s + o * 5, x_out ? x_out + o * M : nullptr, &n_run);
            if (rc == GH_RESIDENT_ABORTED) break;
            if (rc != GH_OK) return rc;
            first_threaded = i + 1;
        }
        if (first_threaded == C) return GH_OK;
    }
    auto work = [&](int i) {
        gh_ctx *k = c->kids[(size_t)i];
        if (hipSetDevice(k->device) != hipSuccess) {
            rcs[(size_t)i] = GH_ERR_HIP;
            return;
        }
        for (int t = 0; t < T; ++t) {
            const size_t src = (size_t)i * T + t, dst = (size_t)i * Tout + t;
            const double *p0 = p0flat ? p0flat + src * M : p0rows[src];
            const double *pn = (t + 1 < T) ? (p0flat ? p0flat + (src + 1) * M : p0rows[src + 1]) : nullptr;
            int acc = 0;
            double o5[5];
            int rc = chain_trajectory_impl(k, p0, dt, L[src], us[src], pn, &acc, o5);
            if (rc == GH_OK && x_out && acc) rc = gh_chain_get_x(k, x_out + dst * M);
            if (rc != GH_OK) {
                rcs[(size_t)i] = rc;
                return;
            }
            accepted[dst] = acc;
            memcpy(out5s + dst * 5, o5, 5 * sizeof(double));
        }
    };
    std::vector<std::thread> pool;
    for (int i = first_threaded + 1; i < C; ++i) pool.emplace_back(work, i);
    work(first_threaded);
    for (std::thread &th : pool) th.join();
    for (int i = 0; i < C; ++i)
        if (rcs[(size_t)i] != GH_OK) return fail(c, rcs[(size_t)i], "chain %d: %s", i, gh_last_error(c->kids[(size_t)i]));
    return GH_OK;
}

int gh_batch_init(gh_ctx *c, int C, const double *x0s, const double *low, const double *high)
{
    if (!c || !x0s || !low || !high) return fail(c, GH_ERR_ARG, "gh_batch_init: null pointer");
    if (C < 1 || C > CB) return fail(c, GH_ERR_ARG, "gh_batch_init: 1..16 chains per batch");
    TRY(need(c, c->have_G && c->have_data && c->have_reg,
             "gh_batch_init: needs the kernel (gh_build_G / gh_upload_G), gh_set_data and gh_set_reg"));
    if (c->sh.kind != 0)
        return fail(c, GH_ERR_UNSUPPORTED, "batched chains run on the unsharded kernel only");
    HIPCHK(c, hipSetDevice(c->device));
    if (lonsym_on(c)) {
        // (the light contexts of the chains share the tables, not a compressed forward operator)
        if (c->wv.on) return fail(c, GH_ERR_UNSUPPORTED, "batched chains on the shift-invariant store run without the wavelet-compressed forward");
        return kids_make(c, C, x0s, low, high);
    }
    // (the wavelet-compressed forward only where the resident chain kernel takes the batch: the MFMA
    // batch has no compressed forward)
    if (c->wv.on && !(resident_usable(c) && resident_lds_doubles(c->ld, c->rs.cpw, C, c->rs.lds_cols, c->rs.split) *
                                                  sizeof(double) <= (size_t)c->rs.lds_max))
        return fail(c, GH_ERR_UNSUPPORTED, "batched chains with the wavelet-compressed forward need a problem "
                                           "small enough for the resident chain kernel");
    TRY(ensure_work(c));
    TRY(h2d(c, c->low, low, (size_t)c->M));
    TRY(h2d(c, c->high, high, (size_t)c->M));
    gh_ctx::Resident &r = c->rs;
    r.b_on = false;
    c->bt.run = gh_ctx::Batch::Run();  // anything gh_batch_run left in flight is discarded
    if (resident_usable(c) &&
        resident_lds_doubles(c->ld, r.cpw, C, r.lds_cols, r.split) * sizeof(double) <= (size_t)r.lds_max) {
        // small problem: the chains take turns inside the resident chain kernel (one launch per
        // round of trajectories, G loaded into LDS once for all of them) -- a sweep of a 30 MB G
        // per launch would leave the MFMA batch bound by launches
        static_assert(CB <= RES_MAX_CHAINS, "chains per batch");
        TRY(dalloc(c, &r.bx, (size_t)CB * (size_t)c->M));
        TRY(dalloc(c, &r.bg, (size_t)CB * (size_t)c->M));
        TRY(dalloc(c, &r.bu, 3 * (size_t)CB));
        TRY(h2d(c, r.bx, x0s, (size_t)C * (size_t)c->M));
        r.b_on = true;
        r.b_state = false;
        c->bt.C = C;
        c->bt.ready = true;
        // two or more chains: all of them in lock-step, one exchange per step of the whole batch (resbatch.hip.h)
        (void)resbatch_plan(c, C);
        return GH_OK;
    }
    return batch_init_mfma(c, C, x0s);
}

int gh_batch_trajectory(gh_ctx *c, const double *p0s, double dt, const int *L, const double *us, int *accepted,
                        double *out5s)
{
    if (!c || !p0s || !L || !us || !accepted || !out5s) return fail(c, GH_ERR_ARG, "gh_batch_trajectory: null pointer");
    gh_ctx::Batch &b = c->bt;
    TRY(need(c, b.ready, "gh_batch_trajectory: call gh_batch_init first"));
    for (int k = 0; k < b.C; ++k)
        if (b.run.live && b.run.active[k])
            return fail(c, GH_ERR_ARG, "gh_batch_trajectory: gh_batch_run left trajectories in flight (drain them with T = 0)");
    b.run.live = false;
    HIPCHK(c, hipSetDevice(c->device));
    const int C = b.C;
    int Lmax = 0;
    for (int k = 0; k < C; ++k) {
        if (L[k] < 1) return fail(c, GH_ERR_ARG, "gh_batch_trajectory: L must be >= 1");
        Lmax = std::max(Lmax, L[k]);
    }
    if (!c->kids.empty()) return kids_run(c, 1, L, nullptr, p0s, us, dt, accepted, out5s, nullptr, 1);
    if (c->rs.b_on && c->rs.ls.on) {
        for (int k = 0; k < C; ++k)
            if (c->rs.ls.active[k])
                return fail(c, GH_ERR_ARG, "gh_batch_trajectory: gh_batch_run left trajectories in flight (drain them with T = 0)");
        int rc = resbatch_state(c);
        if (rc == GH_OK) rc = resbatch_launch(c, 1, L, nullptr, p0s, us, dt, false, accepted, out5s, nullptr, nullptr, nullptr);
        if (rc == GH_OK) return GH_OK;
        if (rc != GH_RESIDENT_ABORTED) return rc;
        c->rs.ls.on = false;  // (nothing was in flight: the chains take turns from here on)
    }
    if (c->rs.b_on) {
        gh_ctx::Resident &r = c->rs;
        int chain_of[CB];
        for (int k = 0; k < C; ++k) chain_of[k] = k;
        ResLaunch q;
        q.C = C;
        q.K = C;
        q.chain_of = chain_of;
        q.L = L;
        q.p0s = p0s;
        q.us = us;
        q.dt = dt;
        q.x_dev = r.bx;
        q.gcur_dev = r.bg;
        q.ucur_dev = r.bu;
        q.have_state = r.b_state ? 1 : 0;
        int h_run[4] = {0, 0, 0, 0};
        const int rc = resident_launch(c, q, accepted, out5s, h_run);
        if (rc == GH_OK) {
            r.b_state = true;
            return GH_OK;
        }
        if (rc != GH_RESIDENT_ABORTED) return rc;
        // the kernel gave up (its workgroups were not all resident): carry on with the MFMA batch
        std::vector<double> xs((size_t)C * (size_t)c->M);
        TRY(d2h(c, xs.data(), r.bx, xs.size()));
        r.b_on = false;
        TRY(batch_init_mfma(c, C, xs.data()));
    }
    const int64_t n16 = c->M * CB;
    TRY(batch_upload_rows(c, p0s, C, b.Pw[0]));
    batch_sumsq_kernel<<<dim3((unsigned)b.n_pp0), dim3(256), 0, c->stream>>>(b.Pw[0], c->M, b.pp0_part);
    const double *X_in = b.Xc, *Rt_in = b.Rtc, *GREG_in = b.GREGc;
    int pin = 0, xo = 0;
    for (int s = 0; s <= Lmax; ++s) {
        BatchAdjArgs a{};
        a.Gb = b.Gb;
        a.G = c->G;
        a.ld = c->ld;
        a.M = c->M;
        a.np = (int)(c->ld / 16);
        a.Rt = Rt_in;
        a.GREG = GREG_in;
        a.X_in = X_in;
        a.P_in = b.Pw[pin];
        a.X_out = b.Xw[xo];
        a.P_out = b.Pw[pin ^ 1];
        a.low = c->low;
        a.high = c->high;
        a.G_out = nullptr;
        a.pp_part = b.pp_part;
        a.dt = dt;
        a.n_waves = b.n_waves;
        bool any_upd = false;
        for (int k = 0; k < CB; ++k) {
            a.phase[k] = PH_IDLE;
            a.cu[k] = (s == 0) ? dt * 0.5 : dt;
            a.cp[k] = dt * 0.5;
            if (k < C) {
                if (s < L[k]) {
                    a.phase[k] = PH_UPD;
                    any_upd = true;
                } else if (s == L[k]) {
                    a.phase[k] = PH_PFIN;
                }
            }
        }
        TRY(batch_launch_adjoint(c, a, any_upd));
        if (any_upd) TRY(batch_evaluate(c, b.Xw[xo], b.Dw, b.GREGw, b.Rtw));
        X_in = b.Xw[xo];
        Rt_in = b.Rtw;
        GREG_in = b.GREGw;
        pin ^= 1;
        xo ^= 1;
    }
    double *h = b.h;
    HIPCHK(c, hipMemcpyAsync(h, b.scal, sizeof(double) * CB * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(h + CB * 4, b.pp_part, sizeof(double) * (size_t)b.n_waves * CB, hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipMemcpyAsync(h + CB * 4 + (size_t)b.n_waves * CB, b.pp0_part, sizeof(double) * (size_t)b.n_pp0 * CB,
                             hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        // a fused team pass of this round gave up: nothing of the chains' current states was touched --
        // run the round again (the fused form is off from here on)
        bool failed = false;
        TRY(mfb_fused_failed(c, &failed));
        if (failed) return gh_batch_trajectory(c, p0s, dt, L, us, accepted, out5s);
    }
    unsigned mask = 0;
    for (int k = 0; k < C; ++k) {
        double pp1 = 0.0, pp0 = 0.0;
        for (int w = 0; w < b.n_waves; ++w) pp1 += h[CB * 4 + (size_t)w * CB + k];
        for (int w = 0; w < b.n_pp0; ++w) pp0 += h[CB * 4 + (size_t)(b.n_waves + w) * CB + k];
        const double Unew[3] = {h[4 * k + 2], h[4 * k + 0], h[4 * k + 1]};
        const double Hcur = 0.5 * pp0 + b.U[k][0];
        const double Hnew = 0.5 * pp1 + Unew[0];
        const bool acc = (Hnew < Hcur) || (us[k] < std::exp(-(Hnew - Hcur)));
        if (acc) {
            mask |= 1u << k;
            b.U[k][0] = Unew[0];
            b.U[k][1] = Unew[1];
            b.U[k][2] = Unew[2];
        }
        accepted[k] = acc ? 1 : 0;
        out5s[5 * k + 0] = b.U[k][0];
        out5s[5 * k + 1] = b.U[k][1];
        out5s[5 * k + 2] = b.U[k][2];
        out5s[5 * k + 3] = Hcur;
        out5s[5 * k + 4] = Hnew;
    }
    if (mask) {
        const int64_t l16 = c->ld * CB;
        batch_commit_kernel<<<dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream>>>(X_in, b.Xc, n16, mask);
        batch_commit_kernel<<<dim3((unsigned)((n16 + 255) / 256)), dim3(256), 0, c->stream>>>(b.GREGw, b.GREGc, n16,
                                                                                            mask);
        batch_commit_kernel<<<dim3((unsigned)((l16 + 255) / 256)), dim3(256), 0, c->stream>>>(b.Dw, b.Dc, l16, mask);
        batch_commit_rt_kernel<<<dim3((unsigned)((l16 + 255) / 256)), dim3(256), 0, c->stream>>>(b.Rtw, b.Rtc, l16,
                                                                                               mask);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipStreamSynchronize(c->stream));
    }
    return GH_OK;
}

int gh_batch_run(gh_ctx *c, int T, const int *L, const double *const *p0s, const double *us, double dt,
                 int *accepted, double *out5s, double *x_out, int *n_started, int *n_done)
{
    if (!c || T < 0 || (T > 0 && (!L || !p0s || !us)) || !accepted || !out5s || ((n_started == nullptr) != (n_done == nullptr)))
        return fail(c, GH_ERR_ARG, "gh_batch_run: bad arguments");
    if (T == 0 && !n_done) return fail(c, GH_ERR_ARG, "gh_batch_run: T = 0 (drain) needs n_started / n_done");
    gh_ctx::Batch &b = c->bt;
    TRY(need(c, b.ready, "gh_batch_run: call gh_batch_init first"));
    HIPCHK(c, hipSetDevice(c->device));
    const int C = b.C;
    const size_t M = (size_t)c->M;
    for (int k = 0; k < C * T; ++k)
        if (L[k] < 1 || !p0s[k]) return fail(c, GH_ERR_ARG, "gh_batch_run: L must be >= 1 and every momentum row given");
    if (!c->kids.empty()) {
        // shift-invariant store: every chain runs its T trajectories on its own stream; nothing stays in flight
        TRY(kids_run(c, T, L, p0s, nullptr, us, dt, accepted, out5s, x_out, n_done ? T + 1 : T));
        for (int ch = 0; ch < C && n_done; ++ch) n_started[ch] = n_done[ch] = T;
        return GH_OK;
    }
    // trajectories the lock-step kernel had in flight when it gave up: replayed in front of the new lists
    std::vector<int> pre_ch, pre_L;
    std::vector<double> pre_us, pre_p0;
    if (c->rs.b_on && c->rs.ls.on) {
        gh_ctx::Resident::LockStep &b = c->rs.ls;
        bool any_active = false;
        for (int ch = 0; ch < C; ++ch) any_active = any_active || b.active[ch];
        if (!n_done && any_active)
            return fail(c, GH_ERR_ARG, "gh_batch_run: trajectories in flight from a carry-over call (drain them with T = 0)");
        if (T == 0 && !any_active) {
            for (int ch = 0; ch < C; ++ch) n_started[ch] = n_done[ch] = 0;
            return GH_OK;
        }
        int rc = resbatch_state(c);
        if (rc == GH_OK)
            rc = resbatch_launch(c, T, L, p0s, nullptr, us, dt, n_done != nullptr, accepted, out5s, x_out, n_started, n_done);
        if (rc == GH_OK) return GH_OK;
        if (rc != GH_RESIDENT_ABORTED) return rc;
        b.on = false;
        TRY(resbatch_inflight(c, pre_ch, pre_L, pre_us, pre_p0));
    }
    if (c->rs.b_on && T == 0 && pre_ch.empty()) {
        for (int ch = 0; ch < C; ++ch) n_started[ch] = n_done[ch] = 0;  // nothing is ever left in flight there
        return GH_OK;
    }
    if (c->rs.b_on) {
        // small problem: the chains take turns inside the resident chain kernel, trajectory t of
        // every chain before trajectory t + 1 of any (in front of them what the lock-step kernel left in flight)
        gh_ctx::Resident &r = c->rs;
        const int P = (int)pre_ch.size(), K = P + C * T;
        std::vector<int> chain_of((size_t)K), Lk((size_t)K), acc((size_t)K), has_pre((size_t)C, 0);
        std::vector<double> pk((size_t)K * M), uk((size_t)K), o5((size_t)K * 5);
        for (int k = 0; k < P; ++k) {
            chain_of[k] = pre_ch[k];
            Lk[k] = pre_L[k];
            uk[k] = pre_us[k];
            memcpy(pk.data() + (size_t)k * M, pre_p0.data() + (size_t)k * M, M * sizeof(double));
            has_pre[(size_t)pre_ch[k]] = 1;
        }
        for (int t = 0; t < T; ++t)
            for (int ch = 0; ch < C; ++ch) {
                const int k = P + t * C + ch, src = ch * T + t;
                chain_of[k] = ch;
                Lk[k] = L[src];
                uk[k] = us[src];
                memcpy(pk.data() + (size_t)k * M, p0s[src], M * sizeof(double));
            }
        ResLaunch q;
        q.C = C;
        q.K = K;
        q.chain_of = chain_of.data();
        q.L = Lk.data();
        q.p0s = pk.data();
        q.us = uk.data();
        q.dt = dt;
        q.x_dev = r.bx;
        q.gcur_dev = r.bg;
        q.ucur_dev = r.bu;
        q.have_state = r.b_state ? 1 : 0;
        q.want_x = x_out != nullptr;
        int h_run[4] = {0, 0, 0, 0};
        const int rc = resident_launch(c, q, acc.data(), o5.data(), h_run);
        if (rc == GH_OK) {
            r.b_state = true;
            const int Tout = n_done ? T + 1 : T;
            for (int k = 0; k < K; ++k) {
                const int ch = chain_of[k];
                const int i = k < P ? 0 : (k - P) / C + has_pre[(size_t)ch];
                const int dst = ch * Tout + i;
                accepted[dst] = acc[k];
                memcpy(out5s + (size_t)dst * 5, o5.data() + (size_t)k * 5, 5 * sizeof(double));
                if (x_out && acc[k])
                    HIPCHK(c, hipMemcpyAsync(x_out + (size_t)dst * M, r.xacc + (size_t)k * M, M * sizeof(double),
                                             hipMemcpyDeviceToHost, c->stream));
            }
            HIPCHK(c, hipStreamSynchronize(c->stream));
            for (int ch = 0; ch < C && n_done; ++ch) {
                n_started[ch] = T;
                n_done[ch] = T + has_pre[(size_t)ch];
            }
            return GH_OK;
        }
        if (rc != GH_RESIDENT_ABORTED) return rc;
        if (P > 0) return fail(c, GH_ERR_HIP, "gh_batch_run: the resident kernels timed out with trajectories in flight");
        std::vector<double> xs((size_t)C * M);
        TRY(d2h(c, xs.data(), r.bx, xs.size()));
        r.b_on = false;
        TRY(batch_init_mfma(c, C, xs.data()));
    }
    // ---- fp64-MFMA batch, chains desynchronised: per sweep every chain is in its own phase.
    // The scheduler's state (b.run) outlives the call when the caller asks for n_started / n_done:
    // the call then ends as soon as a chain has nothing left to start, the others keep their
    // trajectory in flight and carry on in the next call -- no sweep is ever spent waiting for
    // the slowest chain.  Without them every chain's T trajectories are completed.
    const bool carry = n_done != nullptr;
    const bool use_spec = env_int("GRAVHMC_BATCH_SPEC", 1) != 0;
    gh_ctx::Batch::Run &run = b.run;
    const int64_t n16 = c->M * CB, l16 = c->ld * CB;
    const unsigned all = (C >= 32) ? 0xffffffffu : ((1u << C) - 1u);
    auto blocks = [](int64_t n) { return dim3((unsigned)((n + 255) / 256)); };
    double *h = b.h;
    if (!c->copy_stream) {
        HIPCHK(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIPCHK(c, hipEventCreateWithFlags(&c->copy_ev, hipEventDisableTiming));
    }
    TRY(dalloc(c, &b.stage2, (size_t)n16));
    TRY(dalloc(c, &b.GREGw2, (size_t)n16));
    TRY(dalloc(c, &b.Dw2, (size_t)l16));
    TRY(dalloc(c, &b.Rtw2, (size_t)l16));
    TRY(dalloc(c, &b.scal2, CB * 4));
    TRY(dalloc(c, &b.Pn, (size_t)n16));
    TRY(dalloc(c, &b.pn0_part, (size_t)b.n_pp0 * CB));
    // (matrix-free team pass: the momentum every trajectory in flight started with, so that the
    // trajectories can be replayed if a pass gives up)
    const bool keep_pstart = b.fus_on;
    if (keep_pstart) TRY(dalloc(c, &b.Pstart, (size_t)n16));
    // two working sets: a sweep reads set run.ws, the evaluation behind it writes the other one
    double *GREGs[2] = {b.GREGw, b.GREGw2}, *Ds[2] = {b.Dw, b.Dw2}, *Rts[2] = {b.Rtw, b.Rtw2},
           *scals[2] = {b.scal, b.scal2};
    if (!run.live) {
        // working state <- current state of every chain
        run = gh_ctx::Batch::Run();
        batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.Xc, b.Xw[0], n16, all);
        batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.GREGc, GREGs[0], n16, all);
        batch_commit_kernel<<<blocks(l16), dim3(256), 0, c->stream>>>(b.Dc, Ds[0], l16, all);
        batch_commit_rt_kernel<<<blocks(l16), dim3(256), 0, c->stream>>>(b.Rtc, Rts[0], l16, all);
        HIPCHK(c, hipGetLastError());
        run.live = true;
    } else if (run.dt != dt) {
        for (int k = 0; k < C; ++k)
            if (run.active[k]) return fail(c, GH_ERR_ARG, "gh_batch_run: dt changed while trajectories are in flight");
    }
    run.dt = dt;
    std::vector<int> q_of((size_t)C, 0), done_of((size_t)C, 0);
    // The momentum of the trajectory a chain starts next waits in one of the chain's two staging
    // rows (b.stage / b.stage2, used alternately); it is sent on the copy stream while sweeps run.
    std::vector<char> staged((size_t)C, 0);
    auto stage_row = [&](int ch, int par) { return (par ? b.stage2 : b.stage) + (size_t)ch * M; };
    auto upload = [&](int ch, hipStream_t st) -> int {  // list element q_of[ch] -> the chain's free row
        run.par[ch] ^= 1;
        HIPCHK(c, hipMemcpyAsync(stage_row(ch, run.par[ch]), p0s[(size_t)ch * T + q_of[ch]],
                                 M * sizeof(double), hipMemcpyHostToDevice, st));
        staged[ch] = 1;
        return GH_OK;
    };
    // staged rows of the chains in `mask` -> their columns of the interleaved array dst
    auto scatter_staged = [&](unsigned mask, double *dst) {
        unsigned even = 0, odd = 0;
        for (int ch = 0; ch < C; ++ch)
            if (mask & (1u << ch)) (run.par[ch] ? odd : even) |= 1u << ch;
        if (even) batch_scatter_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.stage, c->M, even, dst);
        if (odd) batch_scatter_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.stage2, c->M, odd, dst);
    };
    auto wait_copies = [&]() -> int {  // (momenta sent ahead on the copy stream: wait for its last copy)
        HIPCHK(c, hipEventRecord(c->copy_ev, c->copy_stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->copy_ev, 0));
        return GH_OK;
    };
    std::vector<int> pending;  // chains whose next momentum is still to be sent ahead
    for (int ch = 0; ch < C; ++ch) {
        if (T == 0) break;
        if (!run.active[ch])
            TRY(upload(ch, c->stream));
        else
            pending.push_back(ch);
    }
    // a chain takes the trajectory at the head of its list: bookkeeping shared by both kinds of start
    auto take_next = [&](int ch, double pp0_val, int s_first) {
        const size_t slot = (size_t)ch * T + q_of[ch];
        run.pp0[ch] = pp0_val;
        run.L_cur[ch] = L[slot];
        run.u_cur[ch] = us[slot];
        run.s_of[ch] = s_first;
        run.active[ch] = true;
        q_of[ch] += 1;
        staged[ch] = 0;
        // the one after goes ahead once the next sweep has been queued (the staging copy blocks
        // this thread, not the GPU)
        if (q_of[ch] < T) pending.push_back(ch);
    };
    // working state <- current state, momenta of the chains in `mask` <- their next trajectory
    auto start_chains = [&](unsigned mask) -> int {
        TRY(wait_copies());
        for (int ch = 0; ch < C; ++ch)
            if ((mask & (1u << ch)) && !staged[ch]) TRY(upload(ch, c->stream));
        scatter_staged(mask, b.Pw[run.pin]);
        if (keep_pstart) batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.Pw[run.pin], b.Pstart, n16, mask);
        batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.Xc, b.Xw[run.xi], n16, mask);
        batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.GREGc, GREGs[run.ws], n16, mask);
        batch_commit_kernel<<<blocks(l16), dim3(256), 0, c->stream>>>(b.Dc, Ds[run.ws], l16, mask);
        batch_commit_rt_kernel<<<blocks(l16), dim3(256), 0, c->stream>>>(b.Rtc, Rts[run.ws], l16, mask);
        batch_sumsq_kernel<<<dim3((unsigned)b.n_pp0), dim3(256), 0, c->stream>>>(b.Pw[run.pin], c->M, b.pp0_part);
        HIPCHK(c, hipGetLastError());
        HIPCHK(c, hipMemcpyAsync(h + CB * 4 + (size_t)b.n_waves * CB, b.pp0_part,
                                 sizeof(double) * (size_t)b.n_pp0 * CB, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        for (int ch = 0; ch < C; ++ch)
            if (mask & (1u << ch)) {
                double s = 0.0;
                for (int w = 0; w < b.n_pp0; ++w) s += h[CB * 4 + (size_t)(b.n_waves + w) * CB + ch];
                take_next(ch, s, 0);
            }
        return GH_OK;
    };
    const size_t h_pn0 = CB * 4 + (size_t)(b.n_waves + b.n_pp0) * CB;  // (b.h is sized for it below)
    for (;;) {
        unsigned starters = 0;
        bool starved = false, any_active = false;
        for (int k = 0; k < C; ++k) {
            if (run.active[k]) {
                any_active = true;
            } else if (q_of[k] < T) {
                starters |= 1u << k;
                any_active = true;
            } else {
                starved = true;
            }
        }
        if (carry ? (T == 0 ? !any_active : starved) : !any_active) break;
        if (starters) TRY(start_chains(starters));
        const int rs = run.ws, wset = rs ^ 1;
        BatchAdjArgs a{};
        a.Gb = b.Gb;
        a.G = c->G;
        a.ld = c->ld;
        a.M = c->M;
        a.np = (int)(c->ld / 16);
        a.Rt = Rts[rs];
        a.GREG = GREGs[rs];
        a.X_in = b.Xw[run.xi];
        a.P_in = b.Pw[run.pin];
        a.Pn = b.Pn;
        a.X_out = b.Xw[run.xi ^ 1];
        a.P_out = b.Pw[run.pin ^ 1];
        a.low = c->low;
        a.high = c->high;
        a.G_out = nullptr;
        a.pp_part = b.pp_part;
        a.dt = dt;
        a.n_waves = b.n_waves;
        bool any_upd = false;
        unsigned fin = 0, spec = 0;
        for (int k = 0; k < CB; ++k) {
            a.phase[k] = PH_IDLE;
            a.cu[k] = dt;
            a.cp[k] = dt * 0.5;
            if (k < C && run.active[k]) {
                if (run.s_of[k] < run.L_cur[k]) {
                    a.phase[k] = PH_UPD;
                    a.cu[k] = (run.s_of[k] == 0) ? dt * 0.5 : dt;
                    any_upd = true;
                } else {
                    fin |= 1u << k;
                    // the chain's next trajectory is known: its first step rides on this sweep
                    if (use_spec && q_of[k] < T) {
                        a.phase[k] = PH_PFIN_SPEC;
                        a.cu[k] = dt * 0.5;
                        spec |= 1u << k;
                        any_upd = true;
                    } else {
                        a.phase[k] = PH_PFIN;
                    }
                }
            }
        }
        if (spec) {
            TRY(wait_copies());
            for (int ch = 0; ch < C; ++ch)
                if ((spec & (1u << ch)) && !staged[ch]) TRY(upload(ch, c->stream));
            scatter_staged(spec, b.Pn);
            batch_sumsq_kernel<<<dim3((unsigned)b.n_pp0), dim3(256), 0, c->stream>>>(b.Pn, c->M, b.pn0_part);
        }
        TRY(batch_launch_adjoint(c, a, any_upd));
        if (any_upd) {
            TRY(batch_evaluate(c, b.Xw[run.xi ^ 1], Ds[wset], GREGs[wset], Rts[wset], scals[wset]));
            run.ws = wset;
        }
        for (int ch : pending) TRY(upload(ch, c->copy_stream));
        pending.clear();
        const int x_prop = run.xi;  // the sweep's input: the proposals of the chains that finished
        run.xi ^= 1;
        run.pin ^= 1;
        for (int k = 0; k < C; ++k)
            if (run.active[k]) run.s_of[k] += 1;
        if (!fin) continue;
        // the chains that took their final half step in this sweep: Metropolis test.  Their
        // proposals' potentials, gradients and residuals are in the set the sweep READ.
        HIPCHK(c, hipMemcpyAsync(h, scals[rs], sizeof(double) * CB * 4, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipMemcpyAsync(h + CB * 4, b.pp_part, sizeof(double) * (size_t)b.n_waves * CB,
                                 hipMemcpyDeviceToHost, c->stream));
        if (spec)
            HIPCHK(c, hipMemcpyAsync(h + h_pn0, b.pn0_part, sizeof(double) * (size_t)b.n_pp0 * CB,
                                     hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        {
            // a team pass since the last look gave up: what the trajectories in flight accumulated since is
            // void, the chains' current states are intact (nothing is committed before this point) -- every
            // active chain starts its trajectory again from its current state and its own momentum, on the
            // two-pass kernels (which need no co-residency)
            bool failed = false;
            TRY(mfb_fused_failed(c, &failed));
            if (failed) {
                if (!b.Pstart) return fail(c, GH_ERR_HIP, "gh_batch_run: the fused matrix-free batch pass timed out");
                unsigned act = 0;
                for (int k = 0; k < C; ++k)
                    if (run.active[k]) {
                        act |= 1u << k;
                        run.s_of[k] = 0;
                    }
                batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.Pstart, b.Pw[run.pin], n16, act);
                batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.Xc, b.Xw[run.xi], n16, act);
                batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.GREGc, GREGs[run.ws], n16, act);
                batch_commit_kernel<<<blocks(l16), dim3(256), 0, c->stream>>>(b.Dc, Ds[run.ws], l16, act);
                batch_commit_rt_kernel<<<blocks(l16), dim3(256), 0, c->stream>>>(b.Rtc, Rts[run.ws], l16, act);
                HIPCHK(c, hipGetLastError());
                continue;
            }
        }
        unsigned mask = 0;
        // result slots per chain: T, plus one in carry-over mode for the trajectory that came in flight
        const int Tout = carry ? T + 1 : T;
        for (int k = 0; k < C; ++k) {
            if (!(fin & (1u << k))) continue;
            const size_t slot = (size_t)k * Tout + done_of[k];
            double pp1 = 0.0;
            for (int w = 0; w < b.n_waves; ++w) pp1 += h[CB * 4 + (size_t)w * CB + k];
            const double Unew[3] = {h[4 * k + 2], h[4 * k + 0], h[4 * k + 1]};
            const double Hcur = 0.5 * run.pp0[k] + b.U[k][0];
            const double Hnew = 0.5 * pp1 + Unew[0];
            const bool acc = (Hnew < Hcur) || (run.u_cur[k] < std::exp(-(Hnew - Hcur)));
            if (acc) {
                mask |= 1u << k;
                b.U[k][0] = Unew[0];
                b.U[k][1] = Unew[1];
                b.U[k][2] = Unew[2];
            }
            accepted[slot] = acc ? 1 : 0;
            out5s[5 * slot + 0] = b.U[k][0];
            out5s[5 * slot + 1] = b.U[k][1];
            out5s[5 * slot + 2] = b.U[k][2];
            out5s[5 * slot + 3] = Hcur;
            out5s[5 * slot + 4] = Hnew;
        }
        if (mask) {
            batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.Xw[x_prop], b.Xc, n16, mask);
            batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(GREGs[rs], b.GREGc, n16, mask);
            batch_commit_kernel<<<blocks(l16), dim3(256), 0, c->stream>>>(Ds[rs], b.Dc, l16, mask);
            batch_commit_rt_kernel<<<blocks(l16), dim3(256), 0, c->stream>>>(Rts[rs], b.Rtc, l16, mask);
            HIPCHK(c, hipGetLastError());
            if (x_out)
                for (int k = 0; k < C; ++k)
                    if (mask & (1u << k)) {
                        batch_extract_kernel<<<blocks(c->M), dim3(256), 0, c->stream>>>(b.Xc, k, c->M, c->tmpM);
                        TRY(d2h(c, x_out + ((size_t)k * Tout + done_of[k]) * M, c->tmpM, M));
                    }
        }
        for (int k = 0; k < C; ++k)
            if (fin & (1u << k)) {
                run.active[k] = false;
                done_of[k] += 1;
                if ((spec & mask) & (1u << k)) {
                    // accepted, and the first step of the next trajectory has been taken: carry on
                    double s = 0.0;
                    for (int w = 0; w < b.n_pp0; ++w) s += h[h_pn0 + (size_t)w * CB + k];
                    take_next(k, s, 1);
                    if (keep_pstart) batch_commit_kernel<<<blocks(n16), dim3(256), 0, c->stream>>>(b.Pn, b.Pstart, n16, 1u << k);
                }
            }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->copy_stream));
    bool any_active = false;
    for (int k = 0; k < C; ++k) {
        if (n_done) {
            n_started[k] = q_of[k];
            n_done[k] = done_of[k];
        }
        any_active = any_active || run.active[k];
    }
    if (!any_active) run.live = false;  // (the working buffers are rebuilt from the current state next time)
    return GH_OK;
}

// Diagnostic (not in the public header): accumulated phase times of the shift-invariant pass
// (GRAVHMC_LONSYM_TIMING=1), 100 MHz ticks of workgroup 0.
int gh_debug_lonsym_timing(gh_ctx *c, long long out8[8])
{
    if (!c || !out8) return GH_ERR_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!c->ls || !c->ls->dbg) return GH_OK;
    HIPCHK(c, hipMemcpyAsync(out8, c->ls->dbg, 8 * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

// Diagnostic (not in the public header): accumulated phase times of workgroup 0 of the persistent harmonic pass
// (GRAVHMC_LONSYM_TIMING=1; lonres.hip.h), 100 MHz ticks.
int gh_debug_lonres_timing(gh_ctx *c, long long out16[16])
{
    if (!c || !out16) return GH_ERR_ARG;
    for (int i = 0; i < 16; ++i) out16[i] = 0;
    if (!c->ls || !c->ls->res.dbg) return GH_OK;
    HIPCHK(c, hipMemcpyAsync(out16, c->ls->res.dbg, 16 * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

// Diagnostic (not in the public header): accumulated phase times of the fused matrix-free batch pass
// (GRAVHMC_MFB_TIMING=1), 100 MHz ticks of workgroup (0, 0).
int gh_debug_mfb_timing(gh_ctx *c, long long out8[8])
{
    if (!c || !out8) return GH_ERR_ARG;
    for (int i = 0; i < 8; ++i) out8[i] = 0;
    if (!c->bt.fus_dbg) return GH_OK;
    HIPCHK(c, hipMemcpyAsync(out8, c->bt.fus_dbg, 8 * sizeof(long long), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

int gh_batch_fused_stats(gh_ctx *c, int *members, int *ranges, int64_t *launches, int *timeouts)
{
    if (!c) return GH_ERR_ARG;
    const gh_ctx::Batch &b = c->bt;
    if (members) *members = b.fus_on ? b.fus_members : 0;
    if (ranges) *ranges = b.fus_on ? b.fus_ranges : 0;
    if (launches) *launches = b.fus_launches;
    if (timeouts) *timeouts = b.fus_aborts;
    return GH_OK;
}

int gh_matrix_free_team_stats(gh_ctx *c, int *members, int *ranges, int64_t *launches, int *timeouts)
{
    if (!c) return GH_ERR_ARG;
    const gh_ctx::MfTeam &t = c->mft;
    if (members) *members = t.state == 1 ? t.members : 0;
    if (ranges) *ranges = t.state == 1 ? t.ranges : 0;
    if (launches) *launches = t.launches;
    if (timeouts) *timeouts = t.aborts;
    return GH_OK;
}

int gh_batch_resident_stats(gh_ctx *c, int64_t *launches, int64_t *lock_steps, int64_t *chain_steps, int64_t *lost_steps,
                            int *timeouts)
{
    if (!c) return GH_ERR_ARG;
    const gh_ctx::Resident::LockStep &b = c->rs.ls;
    if (launches) *launches = b.launches;
    if (lock_steps) *lock_steps = b.lock_steps;
    if (chain_steps) *chain_steps = b.chain_steps;
    if (lost_steps) *lost_steps = b.lost;
    if (timeouts) *timeouts = b.aborts;
    return GH_OK;
}

int gh_pinned_alloc(gh_ctx *c, size_t bytes, void **host)
{
    if (!c || !host || bytes == 0) return fail(c, GH_ERR_ARG, "gh_pinned_alloc: null pointer or no bytes");
    HIPCHK(c, hipSetDevice(c->device));
    void *p = nullptr;
    HIPCHK(c, hipHostMalloc(&p, bytes));
    c->pinned.push_back({(char *)p, bytes});
    *host = p;
    return GH_OK;
}

int gh_pinned_free(gh_ctx *c, void *host)
{
    if (!c || !host) return fail(c, GH_ERR_ARG, "gh_pinned_free: null pointer");
    for (size_t i = 0; i < c->pinned.size(); ++i)
        if (c->pinned[i].base == (char *)host) {
            HIPCHK(c, hipSetDevice(c->device));
            if (c->stream) HIPCHK(c, hipStreamSynchronize(c->stream));
            HIPCHK(c, hipHostFree(host));
            c->pinned.erase(c->pinned.begin() + (long)i);
            return GH_OK;
        }
    return fail(c, GH_ERR_ARG, "gh_pinned_free: not a block of gh_pinned_alloc");
}

int gh_batch_staging_stats(gh_ctx *c, int64_t *rows_direct, int64_t *rows_staged)
{
    if (!c) return GH_ERR_ARG;
    if (rows_direct) *rows_direct = c->rs.ls.rows_direct;
    if (rows_staged) *rows_staged = c->rs.ls.rows_staged;
    return GH_OK;
}

int gh_batch_get_x(gh_ctx *c, int chain, double *x)
{
    if (!c || !x) return fail(c, GH_ERR_ARG, "gh_batch_get_x: null pointer");
    TRY(need(c, c->bt.ready && chain >= 0 && chain < c->bt.C, "gh_batch_get_x: no such chain"));
    HIPCHK(c, hipSetDevice(c->device));
    if (!c->kids.empty()) return gh_chain_get_x(c->kids[(size_t)chain], x);
    if (c->rs.b_on) return d2h(c, x, c->rs.bx + (size_t)chain * (size_t)c->M, (size_t)c->M);
    batch_extract_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(c->bt.Xc, chain, c->M,
                                                                                           c->tmpM);
    HIPCHK(c, hipGetLastError());
    return d2h(c, x, c->tmpM, (size_t)c->M);
}

int gh_leapfrog(gh_ctx *c, double *x_inout, const double *p0, double dt, int L, const double *low,
                const double *high, double u, int *accepted, double out5[5], double *dsyn)
{
    TRY(gh_chain_init(c, x_inout, low, high));
    TRY(gh_chain_trajectory(c, p0, dt, L, u, accepted, out5));
    TRY(gh_chain_get_x(c, x_inout));
    if (dsyn) TRY(gh_chain_get_dsyn(c, dsyn));
    return GH_OK;
}

// Diagnostic (not in the public header): time a pure streaming read of the resident G.
int gh_debug_resident_timing(gh_ctx *c, long long out32[32], int64_t *launches, int64_t *evals)
{
    if (!c || !out32) return GH_ERR_ARG;
    if (launches) *launches = c->rs.launches;
    if (evals) *evals = c->rs.evals;
    for (int i = 0; i < 32; ++i) out32[i] = 0;
    HIPCHK(c, hipSetDevice(c->device));
    if (c->rs.ls.dbg && c->rs.ls.launches > 0) {
        // chains in lock-step (resbatch.hip.h): its clocks, launches and lock-steps
        if (launches) *launches = c->rs.ls.launches;
        if (evals) *evals = c->rs.ls.lock_steps;
        HIPCHK(c, hipMemcpy(out32, c->rs.ls.dbg, 32 * sizeof(long long), hipMemcpyDeviceToHost));
        return GH_OK;
    }
    if (!c->rs.dbg) return GH_OK;
    HIPCHK(c, hipMemcpy(out32, c->rs.dbg, 32 * sizeof(long long), hipMemcpyDeviceToHost));
    return GH_OK;
}

int gh_debug_stream_read(gh_ctx *c, int blocks, int threads, int nt, int reps, double *ms_out);

int gh_measure_stream_read(gh_ctx *c, int nt, int reps, double *ms_per_pass)
{
    if (!c || !ms_per_pass || reps < 1) return fail(c, GH_ERR_ARG, "gh_measure_stream_read: bad arguments");
    TRY(need(c, c->have_G && !c->mf, "gh_measure_stream_read: needs a stored kernel matrix"));
    // the best of three launch shapes (the rate of a plain read depends on how many wide loads the
    // chip keeps in flight: 2 x CUs blocks of 256 threads reach ~6.9 TB/s where 2 x CUs of 1024 get 6.5)
    const int shapes[3][2] = {{c->cus * 2, 256}, {c->cus, 512}, {c->cus * 16, 1024}};
    double best = 0.0;
    for (int k = 0; k < 3; ++k) {
        double ms = 0.0;
        TRY(gh_debug_stream_read(c, shapes[k][0], shapes[k][1], nt, reps, &ms));
        if (k == 0 || ms < best) best = ms;
    }
    *ms_per_pass = best;
    return GH_OK;
}

int gh_debug_stream_read(gh_ctx *c, int blocks, int threads, int nt, int reps, double *ms_out)
{
    if (!c || !c->have_G) return GH_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_work(c));
    hipEvent_t e0, e1;
    HIPCHK(c, hipEventCreate(&e0));
    HIPCHK(c, hipEventCreate(&e1));
    const int64_t n2 = c->ld * c->M / 2;
    for (int w = 0; w < 2; ++w) {
        if (w == 1) HIPCHK(c, hipEventRecord(e0, c->stream));
        for (int r = 0; r < (w ? reps : 1); ++r) {
            if (nt)
                stream_read_kernel<true><<<dim3(blocks), dim3(threads), 0, c->stream>>>(c->G, n2, c->tmpN);
            else
                stream_read_kernel<false><<<dim3(blocks), dim3(threads), 0, c->stream>>>(c->G, n2, c->tmpN);
        }
    }
    HIPCHK(c, hipEventRecord(e1, c->stream));
    HIPCHK(c, hipEventSynchronize(e1));
    float t = 0.f;
    HIPCHK(c, hipEventElapsedTime(&t, e0, e1));
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    *ms_out = t / reps;
    return GH_OK;
}

int gh_shard_unique_id(void *id128)
{
    if (!id128) return GH_ERR_ARG;
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(nullptr, GH_ERR_COMM, "%s", err.c_str());
    ncclUniqueId id;
    ncclResult_t r = api->GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, GH_ERR_COMM, "ncclGetUniqueId failed");
    static_assert(sizeof(id) == 128, "ncclUniqueId is 128 bytes");
    memcpy(id128, &id, sizeof id);
    return GH_OK;
}

int gh_shard_init(gh_ctx *c, const void *id128, int rank, int world, int64_t M_global, int64_t m0)
{
    if (!c || !id128) return fail(c, GH_ERR_ARG, "gh_shard_init: null pointer");
    if (c->sh.kind != 0) return fail(c, GH_ERR_ARG, "gh_shard_init: already initialised");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(shard_common_init(c, rank, world, M_global, m0));
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(c, GH_ERR_COMM, "%s", err.c_str());
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t r = api->CommInitRank(&c->sh.comm, world, id, rank);
    if (r != ncclSuccess)
        return fail(c, GH_ERR_COMM, "ncclCommInitRank: %s", api->GetErrorString ? api->GetErrorString(r) : "error");
    c->sh.kind = 1;
    return GH_OK;
}

int gh_shard_init_callback(gh_ctx *c, gh_allreduce_fn fn, void *user, int rank, int world, int64_t M_global,
                           int64_t m0)
{
    if (!c || !fn) return fail(c, GH_ERR_ARG, "gh_shard_init_callback: null pointer");
    if (c->sh.kind != 0) return fail(c, GH_ERR_ARG, "gh_shard_init: already initialised");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(shard_common_init(c, rank, world, M_global, m0));
    c->sh.cb = fn;
    c->sh.user = user;
    c->sh.kind = 2;
    return GH_OK;
}

int gh_shard_init_rows(gh_ctx *c, const void *id128, int rank, int world, int64_t N_global, int64_t n0)
{
    if (!c || !id128) return fail(c, GH_ERR_ARG, "gh_shard_init_rows: null pointer");
    if (c->sh.kind != 0) return fail(c, GH_ERR_ARG, "gh_shard_init_rows: already initialised");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(shard_rows_init(c, rank, world, N_global, n0));
    std::string err;
    RcclApi *api = rccl_api(err);
    if (!api) return fail(c, GH_ERR_COMM, "%s", err.c_str());
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    ncclResult_t r = api->CommInitRank(&c->sh.comm, world, id, rank);
    if (r != ncclSuccess)
        return fail(c, GH_ERR_COMM, "ncclCommInitRank: %s", api->GetErrorString ? api->GetErrorString(r) : "error");
    c->sh.kind = 1;
    return GH_OK;
}

int gh_shard_init_rows_callback(gh_ctx *c, gh_allreduce_fn fn, void *user, int rank, int world, int64_t N_global,
                                int64_t n0)
{
    if (!c || !fn) return fail(c, GH_ERR_ARG, "gh_shard_init_rows_callback: null pointer");
    if (c->sh.kind != 0) return fail(c, GH_ERR_ARG, "gh_shard_init_rows: already initialised");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(shard_rows_init(c, rank, world, N_global, n0));
    c->sh.cb = fn;
    c->sh.user = user;
    c->sh.kind = 2;
    return GH_OK;
}

int gh_shard_allreduce(gh_ctx *c, double *host_buf, int64_t count)
{
    if (!c || !host_buf) return fail(c, GH_ERR_ARG, "gh_shard_allreduce: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    return comm_allreduce_host(c, host_buf, count);
}

// values [0, n) of a row; the row's last value (ends_row) is followed by the newline, every other one by a blank
static int64_t format_fixed8_range(const double *v, int64_t n, char *out, int64_t cap, bool ends_row)
{
    static const char DIG2[] =
        "00010203040506070809101112131415161718192021222324252627282930313233343536373839"
        "40414243444546474849505152535455565758596061626364656667686970717273747576777879"
        "8081828384858687888990919293949596979899";
    int64_t pos = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (cap - pos < 32) return -1;  // (the fast path writes at most 28 bytes; snprintf checks itself)
        const double x = v[i];
        const double a = std::fabs(x);
        bool fast = a < 9.0e15;  // (false for NaN and infinities too)
        uint64_t ip = 0;
        uint32_t fr = 0;
        if (fast) {
            ip = (uint64_t)a;  // floor of a non-negative value
            const double scaled = (a - (double)ip) * 1e8;  // a - floor(a) is exact; the product is within 1 ulp (1.5e-8)
            const double r = (scaled + 6755399441055744.0) - 6755399441055744.0;  // to nearest, ties to even
            // certain only away from the rounding boundary (and from a tie, which printf breaks on
            // the exact binary value)
            if (std::fabs(std::fabs(scaled - r) - 0.5) < 1e-7) fast = false;
            fr = (uint32_t)r;
            if (fr >= 100000000u) {
                fr -= 100000000u;
                ip += 1;
            }
        }
        if (!fast) {
            const int k = snprintf(out + pos, (size_t)(cap - pos), "%.8f", x);
            if (k < 0 || k >= cap - pos - 2) return -1;
            pos += k;
        } else {
            if (std::signbit(x)) out[pos++] = '-';
            if (ip < 10) {
                out[pos++] = (char)('0' + ip);
            } else {
                char tmp[24];
                int nd = 0;
                do {
                    tmp[nd++] = (char)('0' + ip % 10);
                    ip /= 10;
                } while (ip);
                while (nd) out[pos++] = tmp[--nd];
            }
            out[pos++] = '.';
            const uint32_t hi4 = fr / 10000u, lo4 = fr % 10000u;
            const uint32_t d0 = hi4 / 100u, d1 = hi4 % 100u, d2 = lo4 / 100u, d3 = lo4 % 100u;
            memcpy(out + pos, DIG2 + 2 * d0, 2);
            memcpy(out + pos + 2, DIG2 + 2 * d1, 2);
            memcpy(out + pos + 4, DIG2 + 2 * d2, 2);
            memcpy(out + pos + 6, DIG2 + 2 * d3, 2);
            pos += 8;
        }
        out[pos++] = (i + 1 < n || !ends_row) ? ' ' : '\n';
    }
    return pos;
}


int64_t gh_format_row_fixed8(const double *v, int64_t n, char *out, int64_t cap)
{
    if (!v || !out || n < 0) return -1;
    if (n == 0) {
        if (cap < 1) return -1;
        out[0] = '\n';
        return 1;
    }
    // (a model of 72 000 cells is 0.8 ms of formatting per accepted sample -- as long as three of its trajectories on
    // the GPU: long rows are formatted in up to four pieces side by side and joined; the bytes are the same)
    const int T = (int)std::min<int64_t>(4, n / 16384);
    if (T <= 1) return format_fixed8_range(v, n, out, cap, true);
    std::vector<std::vector<char>> buf((size_t)T);
    std::vector<int64_t> len((size_t)T, -1);
    auto work = [&](int t) {
        const int64_t i0 = n * t / T, i1 = n * (t + 1) / T;
        buf[(size_t)t].resize((size_t)(i1 - i0) * 32 + 64);
        len[(size_t)t] = format_fixed8_range(v + i0, i1 - i0, buf[(size_t)t].data(), (int64_t)buf[(size_t)t].size(), t == T - 1);
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(work, t);
    work(0);
    for (std::thread &x : th) x.join();
    int64_t pos = 0;
    for (int t = 0; t < T; ++t) {
        if (len[(size_t)t] < 0 || pos + len[(size_t)t] > cap) return -1;
        memcpy(out + pos, buf[(size_t)t].data(), (size_t)len[(size_t)t]);
        pos += len[(size_t)t];
    }
    return pos;
}

#include "host_rng.h"

int gh_profile_enable(gh_ctx *c, int enable)
{
    if (!c) return GH_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (enable && c->ev.empty()) {
        c->ev.resize(8192);
        for (auto &e : c->ev) HIPCHK(c, hipEventCreate(&e));
        c->ev_bytes.assign(c->ev.size() / 2, 0);
    }
    for (gh_ctx *k : c->kids) TRY(gh_profile_enable(k, enable));
    c->prof = enable != 0;
    c->prof_stride = (c->ld * c->M * 8 < (int64_t)(1 << 30)) ? 16 : 1;
    c->prof_seen = 0;
    c->ev_used = 0;
    c->prof_ms_acc = 0.0;
    c->prof_launches = 0;
    c->prof_res_evals = 0;
    if (enable) {  // (the matrix-free work counters stay readable after profiling is switched off)
        c->mf_launches = 0;
        if (c->mf_stats) HIPCHK(c, hipMemsetAsync(c->mf_stats, 0, sizeof(MfStats), c->stream));
    }
    return GH_OK;
}

int gh_matrix_free_stats(gh_ctx *c, int64_t *entries, int64_t *leaves, int64_t *launches, int64_t *near_entries,
                         int64_t *near_leaves)
{
    if (c && near_entries) *near_entries = c->mf_near_on ? c->mf_near_n : 0;
    if (c && near_leaves) *near_leaves = c->mf_near_on ? c->mf_near_leaves : 0;
    if (!c) return GH_ERR_ARG;
    TRY(need(c, c->mf && c->mf_stats, "gh_matrix_free_stats: context is not matrix-free (or not built)"));
    HIPCHK(c, hipSetDevice(c->device));
    MfStats h{};
    HIPCHK(c, hipMemcpyAsync(&h, c->mf_stats, sizeof h, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (entries) *entries = (int64_t)h.entries;
    if (leaves) *leaves = (int64_t)h.leaves;
    if (launches) *launches = c->mf_launches;
    return GH_OK;
}

int gh_profile_read(gh_ctx *c, double *sweep_ms, int64_t *sweep_launches, int64_t *bytes_per_sweep)
{
    if (!c) return GH_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // The timed launches of the DOMINANT kind only -- those that read the most bytes of G: the
    // one-read sweeps (one workgroup or a team per column, the whole matrix each) where the path has
    // them next to row-panel launches (a panel each), so that time and bytes per launch are those of
    // one kernel and agree with a kernel trace.  (An evaluation inside the resident chain kernel
    // counts as one sweep of the whole matrix.)
    const int64_t full = c->N * c->M * (int64_t)sizeof(double);
    int64_t maxb = c->prof_res_evals > 0 ? full : 0;
    for (size_t i = 0; i < c->ev_used / 2; ++i) maxb = std::max(maxb, c->ev_bytes[i]);
    double ms = (c->prof_res_evals > 0 && maxb == full) ? c->prof_ms_acc : 0.0;
    int64_t timed = (c->prof_res_evals > 0 && maxb == full) ? c->prof_res_evals : 0;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        if (c->ev_bytes[i / 2] != maxb) continue;
        float t = 0.f;
        HIPCHK(c, hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
        ms += t;
        timed += 1;
    }
    for (gh_ctx *k : c->kids) {
        // (a batch on the shift-invariant store: the timed passes of all chains -- they overlap on the GPU, the
        // sum of their durations is not wall time)
        double kms = 0.0;
        int64_t kn = 0, kb = 0;
        TRY(gh_profile_read(k, &kms, &kn, &kb));
        ms += kms;
        timed += kn;
        maxb = std::max(maxb, kb);
    }
    if (sweep_ms) *sweep_ms = ms;
    if (sweep_launches) *sweep_launches = timed;
    if (bytes_per_sweep) *bytes_per_sweep = timed > 0 ? maxb : full;
    return GH_OK;
}

}  // extern "C"
