/*
 * gravhmc_oracle.c -- CPU restatement of the reference's HMC hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under gravinv3dhmc_amd/ (the product) may
 * import, link or call this file.  Allowed users: tests/, __graft_entry__.smoke()
 * (as the checker) and bench.py's cpu_baseline leg.
 *
 * Parity status: PINNED.  oracle/make_golden.py compares every function here with
 * the reference itself (imported from /root/reference in the build container, see
 * oracle/ref_harness.py) and with the values printed in the reference's committed
 * run logs; the resulting vectors are stored under tests/golden/.
 *
 * Each function cites the reference file:line it restates (paths relative to the
 * reference repository root).  Layout convention: every dense matrix is
 * column-major ("Fortran order", one cell = one contiguous column of N
 * observations) with leading dimension ld >= N, which is how the reference ends up
 * holding Aw (inversion/potential.py:259, `A @ WmInv` yields an F-ordered array).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* constants.py:29,34,44 */
static const double ORC_G = 0.00000006673;
static const double ORC_SI2MGAL = 100000.0;
static const double ORC_MEAN_EARTH_RADIUS = 6378137.0;
/* _prism.pyx:21,23 (the literal used by the reference) */
static const double ORC_PI_LIT = 3.1415926535897931159979634685441851615906;

/* ------------------------------------------------------------------ prism */

/* gravmag/_prism.pyx:16-26 */
static inline double safe_atan2(double y, double x)
{
    if (y == 0) return 0;
    if (y > 0 && x < 0) return atan2(y, x) - ORC_PI_LIT;
    if (y < 0 && x < 0) return atan2(y, x) + ORC_PI_LIT;
    return atan2(y, x);
}

/* gravmag/_prism.pyx:28-34 */
static inline double safe_log(double x)
{
    if (x == 0) return 0;
    return log(x);
}

/* gravmag/_prism.pyx:49-50 */
static inline double prism_kernelz(double x, double y, double z, double r)
{
    return -(x * safe_log(y + r) + y * safe_log(x + r) - z * safe_atan2(x * y, z * r));
}

/* One (observation, prism) entry before unit scaling: gravmag/_prism.pyx:272-290.
 * Loop nest k (z) outer, j (y), i (x) inner; index 0 is the UPPER bound (x2,y2,z2). */
static inline double prism_gz_entry(double xp, double yp, double zp, const double *b)
{
    const double X[2] = {b[1], b[0]}, Y[2] = {b[3], b[2]}, Z[2] = {b[5], b[4]};
    double acc = 0.0;
    for (int k = 0; k < 2; ++k) {
        double dz = Z[k] - zp;
        for (int j = 0; j < 2; ++j) {
            double dy = Y[j] - yp;
            for (int i = 0; i < 2; ++i) {
                double dx = X[i] - xp;
                double r = sqrt(dx * dx + dy * dy + dz * dz);
                double kern = prism_kernelz(dx, dy, dz, r);
                double sign = ((i + j + k) & 1) ? -1.0 : 1.0;
                acc += sign * kern;
            }
        }
    }
    return acc;
}

/* Dense prism gz kernel: gravmag/prism.py:291-316 (cell loop, `kernel2d *= G*SI2MGAL`).
 * bounds6: M x 6 row-major (x1,x2,y1,y2,z1,z2), cells already in mesh order with
 * masked cells removed (prism.py:300-301).  K: column-major N x M, leading dim ld. */
ORC_API int orc_prism_gz(int64_t N, const double *xp, const double *yp, const double *zp,
                         int64_t M, const double *bounds6, double *K, int64_t ld)
{
    const double scale = ORC_G * ORC_SI2MGAL;
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < M; ++c) {
        const double *b = bounds6 + 6 * c;
        double *col = K + c * ld;
        for (int64_t l = 0; l < N; ++l)
            col[l] = prism_gz_entry(xp[l], yp[l], zp[l], b) * scale;
    }
    return 0;
}

/* -------------------------------------------------------------- tesseroid */

#define ORC_STACK 100 /* gravmag/tesseroid.py:79 */

/* gravmag/_tesseroid_numba.py:94-111 */
static inline void tess_distance_size(double lon, double coslat, double sinlat, double radius,
                                      double w, double e, double s, double n, double top,
                                      double bottom, double *distance, double *Llon,
                                      double *Llat, double *Lr)
{
    const double d2r = M_PI / 180;
    double rt = 0.5 * (top + bottom) + ORC_MEAN_EARTH_RADIUS;
    double lont = d2r * 0.5 * (w + e);
    double latt = d2r * 0.5 * (s + n);
    double sinlatt = sin(latt), coslatt = cos(latt);
    double cospsi = sinlat * sinlatt + coslat * coslatt * cos(lon - lont);
    *distance = sqrt(radius * radius + rt * rt - 2 * radius * rt * cospsi);
    double rtop = top + ORC_MEAN_EARTH_RADIUS;
    *Llon = rtop * acos(sinlatt * sinlatt + (coslatt * coslatt) * cos(d2r * (e - w)));
    *Llat = rtop * acos(sin(d2r * n) * sin(d2r * s) + cos(d2r * n) * cos(d2r * s));
    *Lr = top - bottom;
}

/* gravmag/_tesseroid_numba.py:135-157 */
static inline int tess_divisions(double distance, double Llon, double Llat, double Lr,
                                 double ratio, int *nlon, int *nlat, int *nr)
{
    int error = 0;
    *nlon = *nlat = *nr = 1;
    if (distance <= ratio * Llon) {
        if (Llon <= 0.1) error = -1; else *nlon = 2;
    }
    if (distance <= ratio * Llat) {
        if (Llat <= 0.1) error = -1; else *nlat = 2;
    }
    if (distance <= ratio * Lr) {
        if (Lr <= 1e3) error = -1; else *nr = 2;
    }
    return error;
}

/* One (observation, tesseroid) entry before unit scaling.
 * Engine: gravmag/_tesseroid_numba.py:32-71; split :114-132; scale_nodes :75-91;
 * kernelz :207-222.  Returns the accumulated error code (0 or -count) through *err;
 * return value -1000 flags stack overflow (the reference raises OverflowError :53-54).
 * *leaves (optional) receives the number of GLQ leaf evaluations. */
static int tess_gz_entry(double lon, double sinlat, double coslat, double radius,
                         const double *bounds, double ratio, double *out, int *err,
                         int64_t *leaves)
{
    static const double nodes[2] = {-0.577350269189625731058868041146,
                                    0.577350269189625731058868041146};
    const double d2r = M_PI / 180;
    double stack[ORC_STACK][6];
    double acc = 0.0;
    int error_code = 0;
    int64_t nleaf = 0;
    for (int i = 0; i < 6; ++i) stack[0][i] = bounds[i];
    int stktop = 0;
    while (stktop >= 0) {
        double w = stack[stktop][0], e = stack[stktop][1], s = stack[stktop][2],
               n = stack[stktop][3], top = stack[stktop][4], bottom = stack[stktop][5];
        stktop -= 1;
        double distance, Llon, Llat, Lr;
        tess_distance_size(lon, coslat, sinlat, radius, w, e, s, n, top, bottom, &distance,
                           &Llon, &Llat, &Lr);
        int nlon, nlat, nr;
        error_code += tess_divisions(distance, Llon, Llat, Lr, ratio, &nlon, &nlat, &nr);
        int new_cells = nlon * nlat * nr;
        if (new_cells > 1) {
            if (new_cells + (stktop + 1) > ORC_STACK) return -1000;
            double dlon = (e - w) / nlon, dlat = (n - s) / nlat, dr = (top - bottom) / nr;
            for (int i = 0; i < nlon; ++i)
                for (int j = 0; j < nlat; ++j)
                    for (int k = 0; k < nr; ++k) {
                        stktop += 1;
                        stack[stktop][0] = w + i * dlon;
                        stack[stktop][1] = w + (i + 1) * dlon;
                        stack[stktop][2] = s + j * dlat;
                        stack[stktop][3] = s + (j + 1) * dlat;
                        stack[stktop][4] = bottom + (k + 1) * dr;
                        stack[stktop][5] = bottom + k * dr;
                    }
        } else {
            /* scale_nodes */
            double lonc[2], sinlatc[2], coslatc[2], rc[2];
            double dlon = d2r * (e - w), dlat = d2r * (n - s), dr = top - bottom;
            for (int i = 0; i < 2; ++i) {
                lonc[i] = 0.5 * dlon * nodes[i] + d2r * 0.5 * (e + w);
                double latc = 0.5 * dlat * nodes[i] + d2r * 0.5 * (n + s);
                sinlatc[i] = sin(latc);
                coslatc[i] = cos(latc);
                rc[i] = (0.5 * dr * nodes[i] + 0.5 * (top + bottom) + ORC_MEAN_EARTH_RADIUS);
            }
            double scale = dlon * dlat * dr * 0.125;
            /* kernelz */
            double r_sqr = radius * radius;
            double result = 0;
            for (int i = 0; i < 2; ++i) {
                double coslon = cos(lon - lonc[i]);
                for (int j = 0; j < 2; ++j) {
                    double cospsi = sinlat * sinlatc[j] + coslat * coslatc[j] * coslon;
                    for (int k = 0; k < 2; ++k) {
                        double l_sqr = r_sqr + rc[k] * rc[k] - 2 * radius * rc[k] * cospsi;
                        double kappa = (rc[k] * rc[k]) * coslatc[j];
                        result += kappa * (rc[k] * cospsi - radius) / pow(l_sqr, 1.5);
                    }
                }
            }
            result *= -1;
            acc += scale * result;
            nleaf += 1;
        }
    }
    *out = acc;
    *err = error_code;
    if (leaves) *leaves = nleaf;
    return 0;
}

/* Dense tesseroid gz kernel: gravmag/tesseroid.py:109-123 (coordinate conversion),
 * :189-232 (cell loop), :421-431 (`kernel2d*SI2MGAL*G`, G = 6.673e-8).
 * bounds6: M x 6 row-major (w,e,s,n,top,bottom) [deg, m].  err_cells: number of cells for
 * which the engine returned a non-zero error code (the reference warns once per such
 * cell, tesseroid.py:228-229).  n_leaves: total GLQ leaf evaluations (diagnostic).
 * Returns 0, or -1000 on stack overflow. */
ORC_API int orc_tess_gz(int64_t N, const double *lon_deg, const double *lat_deg,
                        const double *height, int64_t M, const double *bounds6, double ratio,
                        double *K, int64_t ld, int64_t *err_cells, int64_t *n_leaves)
{
    const double d2r = M_PI / 180;
    double *lon = (double *)malloc(sizeof(double) * 4 * (size_t)(N > 0 ? N : 1));
    double *sinlat = lon + N, *coslat = lon + 2 * N, *radius = lon + 3 * N;
    for (int64_t l = 0; l < N; ++l) {
        /* numpy.radians(x) == x * (pi/180) */
        lon[l] = lon_deg[l] * d2r;
        double lat = lat_deg[l] * d2r;
        sinlat[l] = sin(lat);
        coslat[l] = cos(lat);
        radius[l] = ORC_MEAN_EARTH_RADIUS + height[l];
    }
    int overflow = 0;
    int64_t bad = 0, leaves_total = 0;
#pragma omp parallel for schedule(dynamic, 8) reduction(+ : bad, leaves_total) reduction(| : overflow)
    for (int64_t c = 0; c < M; ++c) {
        const double *b = bounds6 + 6 * c;
        double *col = K + c * ld;
        int cell_err = 0;
        for (int64_t l = 0; l < N; ++l) {
            double v = 0;
            int err = 0;
            int64_t nl = 0;
            if (tess_gz_entry(lon[l], sinlat[l], coslat[l], radius[l], b, ratio, &v, &err, &nl))
                overflow |= 1;
            cell_err += err;
            leaves_total += nl;
            col[l] = v * ORC_SI2MGAL * ORC_G;
        }
        if (cell_err != 0) bad += 1;
    }
    free(lon);
    if (err_cells) *err_cells = bad;
    if (n_leaves) *n_leaves = leaves_total;
    return overflow ? -1000 : 0;
}

/* ------------------------------------------------------ weighting + GEMVs */

/* inversion/potential.py:232-264: wm_j = (sum_i A_ij^2)^weightfactor, Aw = A * diag(1/wm).
 * Scales A in place, writes wm (the diagonal of Wm).  Zero columns: 1/0 = inf in the
 * reference (quirk SURVEY 9.1); here the column is left untouched and wm_j = 0. */
ORC_API void orc_col_weight(int64_t N, int64_t M, double *A, int64_t ld, double weightfactor,
                            double *wm)
{
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < M; ++j) {
        double *col = A + j * ld;
        double ss = 0.0;
        for (int64_t i = 0; i < N; ++i) ss += col[i] * col[i];
        double w = pow(ss, weightfactor);
        wm[j] = w;
        if (w != 0.0) {
            double inv = 1.0 / w;
            for (int64_t i = 0; i < N; ++i) col[i] = col[i] * inv;
        }
    }
}

/* inversion/potential.py:698  dpre = np.dot(Aw, mw) */
ORC_API void orc_forward(int64_t N, int64_t M, const double *A, int64_t ld, const double *x,
                         double *d)
{
#pragma omp parallel
    {
        /* row blocks: every thread owns a row range and walks all columns, so the
         * summation order per row is the serial one whatever the thread count */
        int nt = 1, tid = 0;
#ifdef _OPENMP
        nt = omp_get_num_threads();
        tid = omp_get_thread_num();
#endif
        int64_t chunk = (N + nt - 1) / nt;
        int64_t lo = tid * chunk, hi = lo + chunk > N ? N : lo + chunk;
        for (int64_t i = lo; i < hi; ++i) d[i] = 0.0;
        for (int64_t j = 0; j < M; ++j) {
            const double *col = A + j * ld;
            double xj = x[j];
            for (int64_t i = lo; i < hi; ++i) d[i] += col[i] * xj;
        }
    }
}

/* inversion/potential.py:708  np.dot(Aw.T, r) */
ORC_API void orc_adjoint(int64_t N, int64_t M, const double *A, int64_t ld, const double *r,
                         double *g)
{
#pragma omp parallel for schedule(static)
    for (int64_t j = 0; j < M; ++j) {
        const double *col = A + j * ld;
        double s = 0.0;
        for (int64_t i = 0; i < N; ++i) s += col[i] * r[i];
        g[j] = s;
    }
}

/* CSR sparse forward (gravmag/compressor3D.py:65, compressor1D.py:58: `Gkernelsp @ coeff`) */
ORC_API void orc_csr_matvec(int64_t nrows, const int64_t *indptr, const int32_t *indices,
                            const double *data, const double *x, double *y)
{
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < nrows; ++i) {
        double s = 0.0;
        for (int64_t k = indptr[i]; k < indptr[i + 1]; ++k) s += data[k] * x[indices[k]];
        y[i] = s;
    }
}

/* ---------------------------------------------------------- regularisers */

enum { ORC_DAMPING = 0, ORC_SMOOTHNESS = 1, ORC_MS = 2, ORC_TV = 3 };

/* First-difference operator of inversion/potential.py:266-361 (`fd3d`) applied as a
 * stencil: rows are m[p]-m[p+1] (x), m[p]-m[p+nx] (y), m[p]-m[p+nx*ny] (z); no cell-size
 * weights.  Smoothness :786-796 (value = |Rv|^2, grad = 2 R^T R v);
 * TV :798-810 (value = sum sqrt(t^2+beta), grad = R^T (t / sqrt(t^2+beta))). */
static double reg_fd(int kind, const int *shape, const double *v, double beta, double *grad)
{
    const int64_t nz = shape[0], ny = shape[1], nx = shape[2];
    const int64_t M = nz * ny * nx;
    const int64_t stride[3] = {1, nx, nx * ny};
    double value = 0.0;
    for (int64_t p = 0; p < M; ++p) grad[p] = 0.0;
    for (int64_t k = 0; k < nz; ++k)
        for (int64_t j = 0; j < ny; ++j)
            for (int64_t i = 0; i < nx; ++i) {
                int64_t p = (k * ny + j) * nx + i;
                const int has[3] = {i < nx - 1, j < ny - 1, k < nz - 1};
                for (int a = 0; a < 3; ++a) {
                    if (!has[a]) continue;
                    int64_t q = p + stride[a];
                    double t = v[p] - v[q];
                    double gq;
                    if (kind == ORC_SMOOTHNESS) {
                        value += t * t;
                        gq = 2.0 * t;
                    } else {
                        double s = sqrt(t * t + beta);
                        value += s;
                        gq = t / s;
                    }
                    grad[p] += gq;
                    grad[q] -= gq;
                }
            }
    return value;
}

/* Regulariser value and gradient.
 * Damping: inversion/potential.py:775-784.  MS: :719-736 (wm2 = diag(WmSquare) = wm^2).
 * shape = (nz, ny, nx) is used by Smoothness/TV only and must satisfy nz*ny*nx == M
 * (the reference raises on carved meshes). */
ORC_API double orc_regulariser(int kind, int64_t M, const int *shape, const double *mw,
                               const double *mwapr, const double *wm2, double beta,
                               double *grad)
{
    double value = 0.0;
    if (kind == ORC_DAMPING) {
        for (int64_t j = 0; j < M; ++j) {
            double v = mw[j] - mwapr[j];
            value += v * v;
            grad[j] = 2 * v;
        }
    } else if (kind == ORC_MS) {
        for (int64_t j = 0; j < M; ++j) {
            double v = mw[j] - mwapr[j];
            double v2 = v * v;
            double den = v2 + beta;
            value += (wm2[j] * v2) / den;
            grad[j] = (2 * beta * wm2[j] * v) / (den * den);
        }
    } else {
        double *v = (double *)malloc(sizeof(double) * (size_t)M);
        for (int64_t j = 0; j < M; ++j) v[j] = mw[j] - mwapr[j];
        value = reg_fd(kind, shape, v, beta, grad);
        free(v);
    }
    return value;
}

/* --------------------------------------------------- potential and gradient */

typedef struct {
    int64_t N, M, ld;
    const double *Aw;       /* column-major N x M, unit-norm columns */
    const double *dobs;     /* N */
    const double *grav_fix; /* N or NULL (potential.py:700-703) */
    const double *wm2;      /* M, column-norm^2 (MS) or NULL */
    const double *mwapr;    /* M */
    int reg_kind;
    int shape[3];
    double alpha, beta;
    /* optional sparse forward (wavelet path): d = csr @ (W x); NULL => dense */
    const int64_t *csr_indptr;
    const int32_t *csr_indices;
    const double *csr_data;
    int64_t csr_ncols;
    int wavelet_dims; /* 0 none, 1, 3 -- the DWT itself is done by the caller hook */
    void (*dwt)(const double *x, double *coeff, void *user);
    void *dwt_user;
} orc_problem;

/* inversion/potential.py:688-717 (data_all) + :812-845 (misfit_and_grad), 'mandatory'
 * constraint (mw = x).  out3 = (misfit, data_value, model_value).  dpre excludes
 * grav_fix.  work: scratch of N + M doubles. */
ORC_API void orc_misfit_and_grad(const orc_problem *P, const double *x, double *out3,
                                 double *grad, double *dpre, double *work)
{
    const int64_t N = P->N, M = P->M;
    double *r = work, *mgrad = work + N;
    if (P->csr_data && P->dwt) {
        double *coeff = (double *)malloc(sizeof(double) * (size_t)P->csr_ncols);
        P->dwt(x, coeff, P->dwt_user);
        orc_csr_matvec(N, P->csr_indptr, P->csr_indices, P->csr_data, coeff, dpre);
        free(coeff);
    } else {
        orc_forward(N, M, P->Aw, P->ld, x, dpre);
    }
    double sd = 0.0, so = 0.0;
    for (int64_t i = 0; i < N; ++i) {
        double di = dpre[i] + (P->grav_fix ? P->grav_fix[i] : 0.0);
        r[i] = di;
        sd += di;
        so += P->dobs[i];
    }
    double md = sd / (double)N, mo = so / (double)N;
    double dv = 0.0;
    for (int64_t i = 0; i < N; ++i) {
        r[i] = (r[i] - md) - (P->dobs[i] - mo);
        dv += r[i] * r[i];
    }
    orc_adjoint(N, M, P->Aw, P->ld, r, grad);
    double mv = orc_regulariser(P->reg_kind, M, P->shape, x, P->mwapr, P->wm2, P->beta, mgrad);
    for (int64_t j = 0; j < M; ++j) grad[j] = 2 * grad[j] + P->alpha * mgrad[j];
    out3[0] = dv + P->alpha * mv;
    out3[1] = dv;
    out3[2] = mv;
}

/* ------------------------------------------------------------- trajectory */

/* One HMC trajectory: inversion/hmc.py:85-177 with the 'mandatory' clamp-and-reflect
 * bounds (:121-144) and identity mass (:44-50).  p0 = randn(M)*Sigma and u = rand() are
 * drawn by the caller in the reference's RNG order (hmc.py:297,95,164).
 * x is updated in place when the proposal is accepted.
 * out = (U, U_data, U_model, Hcur, Hnew) where U.. belong to the returned state
 * (the proposal if accepted, else the starting point, hmc.py:165-173).
 * dsyn (N, may be NULL) receives the matching dpre.  Returns accepted flag (0/1).
 * n_evals (may be NULL) counts misfit_and_grad calls (L+1). */
ORC_API int orc_leapfrog(const orc_problem *P, double *x, const double *p0, double dt, int L,
                         const double *low, const double *high, double u, double *out,
                         double *dsyn, int64_t *n_evals)
{
    const int64_t N = P->N, M = P->M;
    double *buf = (double *)malloc(sizeof(double) * (size_t)(4 * M + 3 * N + N + M));
    double *xn = buf, *pn = xn + M, *grad = pn + M, *d0 = grad + M, *dn = d0 + N,
           *work = dn + N; /* N + M */
    double o0[3], o1[3] = {0, 0, 0};
    for (int64_t j = 0; j < M; ++j) {
        xn[j] = x[j];
        pn[j] = p0[j];
    }
    double K = 0.0;
    for (int64_t j = 0; j < M; ++j) K += pn[j] * pn[j];
    K *= 0.5;
    orc_misfit_and_grad(P, xn, o0, grad, d0, work);
    double Hcur = K + o0[0];
    for (int64_t j = 0; j < M; ++j) pn[j] -= dt * grad[j] * 0.5;
    for (int i = 0; i < L; ++i) {
        for (int64_t j = 0; j < M; ++j) {
            xn[j] += dt * pn[j];
            if (xn[j] > high[j]) {
                xn[j] = high[j];
                pn[j] = -pn[j];
            } else if (xn[j] < low[j]) {
                xn[j] = low[j];
                pn[j] = -pn[j];
            }
        }
        orc_misfit_and_grad(P, xn, o1, grad, dn, work);
        if (i < L - 1)
            for (int64_t j = 0; j < M; ++j) pn[j] -= dt * grad[j];
        else
            for (int64_t j = 0; j < M; ++j) pn[j] -= dt * grad[j] * 0.5;
    }
    double Kn = 0.0;
    for (int64_t j = 0; j < M; ++j) Kn += pn[j] * pn[j]; /* p = -p leaves p.p unchanged */
    Kn *= 0.5;
    double Hnew = Kn + o1[0];
    int accept = (Hnew < Hcur) || (u < exp(-(Hnew - Hcur)));
    const double *o = accept ? o1 : o0;
    if (accept)
        for (int64_t j = 0; j < M; ++j) x[j] = xn[j];
    out[0] = o[0];
    out[1] = o[1];
    out[2] = o[2];
    out[3] = Hcur;
    out[4] = Hnew;
    if (dsyn) memcpy(dsyn, accept ? dn : d0, sizeof(double) * (size_t)N);
    if (n_evals) *n_evals = L + 1;
    free(buf);
    return accept;
}

ORC_API int orc_sizeof_problem(void) { return (int)sizeof(orc_problem); }

/* test-infrastructure knob: the size of the OpenMP teams (oracle.py passes the cores the cgroup really grants) */
ORC_API void orc_set_threads(int n) { omp_set_num_threads(n > 0 ? n : 1); }
