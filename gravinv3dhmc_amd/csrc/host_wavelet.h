// libgravhmc host side: wavelet-compressed forward operator (DWT plan and passes, CSR product,
// dense model-space form).  Included once by gravhmc.hip.
#pragma once

// ------------------------------------------------------------------ wavelet forward

static void wavelet_plan(gh_ctx::Wavelet &w)
{
    for (int k = 0; k < 3; ++k) w.X[0][k] = w.shape[k];
    for (int i = 1; i <= w.levels; ++i)
        for (int k = 0; k < 3; ++k) w.X[i][k] = w.tax[k] ? (w.X[i - 1][k] + 1) / 2 : w.X[i - 1][k];
    int a[3];
    for (int k = 0; k < 3; ++k) a[k] = w.X[w.levels][k];
    for (int i = w.levels; i >= 1; --i)
        for (int k = 0; k < 3; ++k) {
            w.offd[i][k] = a[k];
            if (w.tax[k]) a[k] += w.X[i][k];
        }
    for (int k = 0; k < 3; ++k) w.D[k] = a[k];
    w.Mp = (int64_t)a[0] * a[1] * a[2];
}

// Passes of the one-launch transform (dwt_lds_kernel): the same sequence run_dwt launches, every
// block dense in LDS.  lds_bytes = 0 when a block exceeds the LDS or a thread's register budget.
static void wavelet_plan_lds(gh_ctx *c)
{
    gh_ctx::Wavelet &w = c->wv;
    DwtLdsPlan &pl = w.lds_plan;
    pl = DwtLdsPlan();
    w.lds_bytes = 0;
    // One workgroup does what the pass-per-launch form spreads over the chip: it wins only while the
    // launches cost more than the arithmetic.  Measured (sweep path, per potential evaluation):
    // 19 x 30 x 30 = 17100 cells (ratiogrid): +52 us (one CU needs ~46 us for the 7*10^6 lane
    // operations of the six passes); 10 x 30 x 20 = 6000 cells (C3 on the sweep path): +7 us.  Used for blocks <= GRAVHMC_DWT_LDS_MAX = 2048 doubles.
    if (env_int("GRAVHMC_DWT_LDS", 1) == 0) return;
    pl.M = c->M;
    pl.Cs[0] = (int64_t)w.D[1] * w.D[2];
    pl.Cs[1] = w.D[2];
    pl.Cs[2] = 1;
    int axes[3], na = 0;
    for (int k = 0; k < 3; ++k)
        if (w.tax[k]) axes[na++] = k;
    int64_t maxvol = c->M;
    int np = 0;
    for (int lev = 1; lev <= w.levels; ++lev) {
        int e[3] = {w.X[lev - 1][0], w.X[lev - 1][1], w.X[lev - 1][2]};
        for (int p = 0; p < na; ++p) {
            if (np >= DWT_LDS_MAXPASS) return;
            const int ax = axes[p];
            DwtLdsPass &P = pl.p[np++];
            for (int k = 0; k < 3; ++k) P.e[k] = e[k];
            P.axis = ax;
            P.last = (p == na - 1) ? 1 : 0;
            for (int k = 0; k < 3; ++k) {
                P.split[k] = w.tax[k] ? w.X[lev][k] : 0x7fffffff;
                P.off1[k] = w.tax[k] ? w.offd[lev][k] : 0;
            }
            const int h = (e[ax] + 1) / 2;
            e[ax] = 2 * h;
            const int64_t vol = (int64_t)e[0] * e[1] * e[2];
            maxvol = std::max(maxvol, vol);
            if (vol / 2 > (int64_t)DWT_LDS_NP * DWT_LDS_THREADS) return;
        }
    }
    pl.npass = np;
    int lds_max = 0;
    if (hipDeviceGetAttribute(&lds_max, hipDeviceAttributeMaxSharedMemoryPerBlock, c->device) != hipSuccess) return;
    const size_t need = (size_t)maxvol * sizeof(double);
    if (need > (size_t)lds_max || maxvol > env_int("GRAVHMC_DWT_LDS_MAX", 2048)) return;
    if (allow_dynamic_lds(reinterpret_cast<const void *>(dwt_lds_kernel), need) != hipSuccess) {
        (void)hipGetLastError();
        return;
    }
    w.lds_bytes = need;
}

// Multi-level DWT of `batch` model-shaped vectors x (batch stride xb) into the packed
// coefficient layout C (batch stride Mp, must be zero-initialised: odd lengths leave gaps).
static int run_dwt(gh_ctx *c, const double *x, int64_t xb, int64_t batch, double *C, double *S1,
                   double *S2)
{
    const gh_ctx::Wavelet &w = c->wv;
    const int64_t Cs[3] = {(int64_t)w.D[1] * w.D[2], (int64_t)w.D[2], 1};
    int axes[3], na = 0;
    for (int k = 0; k < 3; ++k)
        if (w.tax[k]) axes[na++] = k;
    for (int lev = 1; lev <= w.levels; ++lev) {
        const double *src = (lev == 1) ? x : C;
        int64_t src_b = (lev == 1) ? xb : w.Mp;
        int e[3] = {w.X[lev - 1][0], w.X[lev - 1][1], w.X[lev - 1][2]};
        int64_t ss[3];
        if (lev == 1) {
            ss[0] = (int64_t)e[1] * e[2];
            ss[1] = e[2];
            ss[2] = 1;
        } else {
            ss[0] = Cs[0];
            ss[1] = Cs[1];
            ss[2] = Cs[2];
        }
        if (na == 1 && lev > 1) {
            // single pass reading and writing C would overlap: stage the input block
            const int64_t len = (int64_t)e[0] * e[1] * e[2];  // contiguous: only the last axis varies
            for (int64_t b = 0; b < batch; ++b)
                HIPCHK(c, hipMemcpyAsync(S1 + b * w.Mp, C + b * w.Mp, len * sizeof(double),
                                         hipMemcpyDeviceToDevice, c->stream));
            src = S1;
            ss[0] = (int64_t)e[1] * e[2];
            ss[1] = e[2];
            ss[2] = 1;
        }
        for (int p = 0; p < na; ++p) {
            const int ax = axes[p];
            const bool last = (p == na - 1);
            double *dst = last ? C : ((p & 1) ? S2 : S1);
            if (!last && dst == src) dst = (dst == S1) ? S2 : S1;
            DwtArgs a{};
            a.in = src;
            a.out = dst;
            a.batch = batch;
            a.in_bstride = src_b;
            a.out_bstride = w.Mp;
            a.axis = ax;
            const int h = (e[ax] + 1) / 2;
            int oe[3] = {e[0], e[1], e[2]};
            oe[ax] = 2 * h;
            for (int k = 0; k < 3; ++k) {
                a.e[k] = e[k];
                a.in_s[k] = ss[k];
                a.in_off[k][0] = a.in_off[k][1] = 0;
                a.in_split[k] = 0x7fffffff;
            }
            if (last) {
                for (int k = 0; k < 3; ++k) {
                    a.out_s[k] = Cs[k];
                    a.out_off[k][0] = 0;
                    if (w.tax[k]) {
                        a.out_split[k] = w.X[lev][k];
                        a.out_off[k][1] = w.offd[lev][k];
                    } else {
                        a.out_split[k] = 0x7fffffff;
                        a.out_off[k][1] = 0;
                    }
                }
            } else {
                a.out_s[0] = (int64_t)oe[1] * oe[2];
                a.out_s[1] = oe[2];
                a.out_s[2] = 1;
                for (int k = 0; k < 3; ++k) {
                    a.out_off[k][0] = 0;
                    a.out_off[k][1] = (k == ax) ? h : 0;
                    a.out_split[k] = 0x7fffffff;
                }
            }
            const int64_t total = (int64_t)oe[0] * oe[1] * oe[2] / 2 * batch;
            const int64_t blocks = std::min<int64_t>((total + 255) / 256, 1 << 20);
            dwt_axis_kernel<<<dim3((unsigned)std::max<int64_t>(blocks, 1)), dim3(256), 0, c->stream>>>(a);
            // the next pass reads what this one wrote: dense block of extents oe
            src = dst;
            src_b = w.Mp;
            e[0] = oe[0];
            e[1] = oe[1];
            e[2] = oe[2];
            ss[0] = a.out_s[0];
            ss[1] = a.out_s[1];
            ss[2] = a.out_s[2];
        }
    }
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

// d = Awcp @ W(x): compressor3D.py:47-68 / compressor1D.py:45-60
static int wavelet_forward(gh_ctx *c, const double *x, double *d_out)
{
    gh_ctx::Wavelet &w = c->wv;
    if (w.lds_bytes) {
        // (w.coeff was zeroed when it was allocated; the transform never writes the packing's gaps)
        dwt_lds_kernel<<<dim3(1), dim3(DWT_LDS_THREADS), w.lds_bytes, c->stream>>>(w.lds_plan, x, w.coeff);
    } else {
        HIPCHK(c, hipMemsetAsync(w.coeff, 0, sizeof(double) * (size_t)w.Mp, c->stream));
        TRY(run_dwt(c, x, c->M, 1, w.coeff, w.s1, w.s2));
    }
    spmv_kernel<<<dim3((unsigned)((c->ld + 3) / 4)), dim3(256), 0, c->stream>>>(
        w.indptr, w.indices, w.data, w.coeff, c->N, c->ld, d_out);
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

// F = Awcp W as a dense N x M matrix: column j is the compressed forward of the unit model e_j
// (exactly the operator the reference applies, thresholding included; only the association of the
// sums differs from DWT-then-SpMV).  For problems small enough for the resident chain kernel, which
// keeps it in LDS: 64 unit vectors per batch of DWT passes + one batched SpMV.
static int wavelet_dense_form(gh_ctx *c)
{
    gh_ctx::Wavelet &w = c->wv;
    if (w.F_valid) return GH_OK;
    const int64_t M = c->M, Mp = w.Mp, B = 64;
    TRY(dalloc(c, &w.F, (size_t)c->ld * (size_t)M));
    double *X = nullptr, *C = nullptr, *S1 = nullptr, *S2 = nullptr;
    HIPCHK(c, hipMalloc((void **)&X, sizeof(double) * (size_t)(B * M)));
    HIPCHK(c, hipMalloc((void **)&C, sizeof(double) * (size_t)(B * Mp)));
    HIPCHK(c, hipMalloc((void **)&S1, sizeof(double) * (size_t)(B * Mp)));
    HIPCHK(c, hipMalloc((void **)&S2, sizeof(double) * (size_t)(B * Mp)));
    int rc = GH_OK;
    for (int64_t j0 = 0; j0 < M && rc == GH_OK; j0 += B) {
        const int64_t nb = std::min(B, M - j0);
        unit_rows_kernel<<<dim3((unsigned)std::min<int64_t>(1024, (nb * M + 255) / 256)), dim3(256), 0, c->stream>>>(
            X, M, j0, nb);
        hipMemsetAsync(C, 0, sizeof(double) * (size_t)(nb * Mp), c->stream);
        rc = run_dwt(c, X, M, nb, C, S1, S2);
        if (rc != GH_OK) break;
        spmv_kernel<<<dim3((unsigned)((c->ld + 3) / 4), (unsigned)nb), dim3(256), 0, c->stream>>>(
            w.indptr, w.indices, w.data, C, c->N, c->ld, w.F + j0 * c->ld, Mp, c->ld);
    }
    hipError_t e = hipStreamSynchronize(c->stream);
    hipFree(X);
    hipFree(C);
    hipFree(S1);
    hipFree(S2);
    if (rc != GH_OK) return rc;
    if (e != hipSuccess) return fail(c, GH_ERR_HIP, "wavelet_dense_form: %s", hipGetErrorString(e));
    w.F_valid = true;
    return GH_OK;
}
