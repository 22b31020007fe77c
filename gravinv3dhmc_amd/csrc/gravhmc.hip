// libgravhmc: host side of the C-ABI declared in include/gravhmc.h (HIP, gfx950 only).
// One context = one GPU + one stream + one inversion problem resident in HBM.
#include "../../include/gravhmc.h"
#include "kernels.hip.h"
#include "batch.hip.h"
#include "resident.hip.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace ghk;

static thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct gh_ctx {
    int device = 0;
    int64_t N = 0, M = 0, ld = 0;
    int cus = 0;
    hipStream_t stream = nullptr;
    std::string err;
    std::vector<void *> allocs;

    // geometry / kernel
    double *obs[3] = {nullptr, nullptr, nullptr};
    double *bounds = nullptr;
    int cell_kind = -1;
    double ratio = 1.6;
    bool have_obs = false, have_cells = false, have_G = false, weighted = false;
    double *G = nullptr;
    int64_t warn_cells = 0, leaves = 0;
    bool mf = false;          // matrix-free: entries are re-evaluated, G is never stored
    bool dense_ok = true;     // N fits the register-resident sweep (<= 16384 rows)
    double *tconv = nullptr;  // tesseroid obs converted to (lon rad, sin lat, cos lat, radius)
    int64_t mf_cells_per_chunk = 0;

    // sweep configuration
    int TW = 0, EPT2 = 0, PF = 1;
    int n_panels = 1;        // row panels of the dense sweep (N > 16384 rows: > 1, two reads of G per step)
    int64_t panel_rows = 0;
    double *gbuf = nullptr;   // gradient accumulated over the panels
    bool NT = false;
    int n_teams = 0, grid = 0;
    int n_teams_sweep = 0;   // teams of the sweep launch (n_teams may be larger: size of the pp partials)
    int64_t cols_per_team = 0;
    size_t lds_bytes = 0;

    // problem vectors
    double *dobs_c = nullptr, *gfix = nullptr, *mwapr = nullptr, *wm = nullptr, *wm2 = nullptr;
    double *low = nullptr, *high = nullptr;
    bool have_data = false, have_fix = false, have_reg = false;
    int reg_kind = 0, shape[3] = {1, 1, 1};
    double alpha = 1.0, beta = 0.01;

    // chain state: three (r, greg, d, scal) sets and three x buffers rotate between "current
    // sample", "proposal" and "speculative first step of the next trajectory"; set/buffer 3 is
    // private to gh_misfit_and_grad.  Swapping indices makes accept/reject free.
    struct StateSet {
        double *r = nullptr, *greg = nullptr, *d = nullptr, *scal = nullptr;
    } st[4];
    double *xb[4] = {nullptr, nullptr, nullptr, nullptr};
    double *pb[2] = {nullptr, nullptr};
    double *pn = nullptr;  // momentum of the NEXT trajectory (gh_chain_prefetch_momentum)
    int cur = 0, xcur = 0;
    bool pn_valid = false, spec_valid = false;
    double pn_probe[3] = {0, 0, 0}, spec_probe[3] = {0, 0, 0};
    double spec_dt = 0.0, spec_pp0 = 0.0, pn_pp0 = 0.0, spec_U[3] = {0, 0, 0};
    int spec_set = 0, spec_x = 0, spec_p = 0;
    int64_t spec_hits = 0, spec_misses = 0, accept_count = 0;
    double *slab2 = nullptr;
    int slab2_rows = 0;
    double *slab = nullptr, *dpart = nullptr, *regpart = nullptr, *pp_part = nullptr,
           *ppn_part = nullptr, *pp0_part = nullptr, *scal_all = nullptr;
    double *tmpM = nullptr, *tmpN = nullptr;
    int n_dpart = 0, n_regpart = 0, n_pp0 = 0;
    double *h_scal = nullptr;  // pinned: scalars + partial sums
    size_t h_scal_n = 0;
    bool chain_ready = false;
    double U_cur[3] = {0, 0, 0};

    // column-block sharding of ONE chain over several GPUs: this context holds
    // the cells [m0, m0 + M) of M_global; N-vectors are replicated, the forward partials are
    // summed across ranks once per potential evaluation.
    struct Shard {
        int kind = 0;  // 0 single GPU, 1 RCCL all-reduce on the stream, 2 host callback
        int rank = 0, world = 1;
        int64_t M_global = 0, m0 = 0;
        ncclComm_t comm = nullptr;
        gh_allreduce_fn cb = nullptr;
        void *user = nullptr;
        double *buf = nullptr;    // device: [d partial (ld) | R partial | pad | boundary planes (halo)]
        double *hbuf = nullptr;   // pinned staging for the callback path
        size_t buf_n = 0;         // doubles in buf / hbuf
        // Smoothness / TV on cells sharded in whole z-planes: the ranks exchange their boundary
        // planes of the model once per evaluation (inside the forward partial's all-reduce)
        bool halo = false;
        int64_t P = 0;            // cells per plane (ny * nx)
        double *alo = nullptr, *ahi = nullptr;  // prior model of the planes below / above
        double *rb = nullptr;     // regulariser partial for its own (2-double) all-reduce
        int64_t collectives = 0;
    } sh;

    // wavelet-compressed forward operator (compressor1D/3D): CSR N x Mp on the device
    struct Wavelet {
        bool on = false;
        int dims = 0, levels = 2;
        int shape[3] = {1, 1, 1};
        int X[5][3];      // X[i]: extents of the blocks level i produces (X[0] = model shape)
        int offd[5][3];   // packed offset of level i's detail pieces (pywt.coeffs_to_array)
        int D[3] = {1, 1, 1};
        bool tax[3] = {false, false, true};  // transformed axes
        int64_t Mp = 0, nnz = 0;
        double thr = 1e-3;
        int64_t *indptr = nullptr;
        int *indices = nullptr;
        double *data = nullptr;
        double *coeff = nullptr, *s1 = nullptr, *s2 = nullptr;  // model-sized scratch
        double *F = nullptr;  // dense model-space form Awcp W (ld x M, column-major), built on demand
        bool F_valid = false;
    } wv;

    // several chains sharing every sweep of G (fp64 MFMA path, batch.hip.h)
    struct Batch {
        int C = 0;
        double *Xc = nullptr, *Rtc = nullptr, *GREGc = nullptr, *Dc = nullptr;   // current states
        double *Xw[2] = {nullptr, nullptr}, *Pw[2] = {nullptr, nullptr};
        double *Rtw = nullptr, *GREGw = nullptr, *Dw = nullptr, *scal = nullptr;
        double *slab = nullptr, *regpart = nullptr, *pp_part = nullptr, *pp0_part = nullptr;
        double *stage = nullptr;  // C x M rows as the host passes them
        double *Gb = nullptr;     // second copy of G in MFMA operand order (adjoint), if HBM allows
        double *h = nullptr;      // pinned
        int n_colblocks = 0, n_regblocks = 0, n_waves = 0, n_pp0 = 0;
        int64_t cols_per_block = 0;
        double U[CB][3];
        bool ready = false;
        int64_t sweeps = 0;
    } bt;

    // resident chain kernel (resident.hip.h): G held in LDS across a whole batch of trajectories
    struct Resident {
        int state = 0;  // 0 not planned yet, 1 usable, -1 not applicable
        int cpw = 0, nwg = 0, rc = 0, ct = 0;  // ct: columns per wave kept in registers
        size_t lds = 0;
        ghk::u64 *slabg = nullptr, *xslabg = nullptr, *dclg = nullptr, *scalg = nullptr, *xccg = nullptr;
        double *xpub = nullptr;
        unsigned *abort_w = nullptr;
        unsigned tag = 0, tagE = 0;  // granule tags used so far (the buffers keep them across launches)
        int Kcap = 0;
        int *L = nullptr, *accepted = nullptr, *n_run = nullptr, *chain = nullptr;
        int lds_max = 0;
        // several chains sharing the resident G (gh_batch_* on small problems)
        double *bx = nullptr, *bg = nullptr, *bu = nullptr;  // C x M models, C x M gradients, 3 C potentials
        bool b_on = false, b_state = false;
        double *p0s = nullptr, *us = nullptr, *out5s = nullptr, *xacc = nullptr;
        int64_t launches = 0, evals = 0;
        long long *dbg = nullptr;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
    } rs;
    int64_t prof_res_evals = 0;

    // ring of the last K accepted samples (posterior statistics without text I/O)
    double *ring = nullptr, *ring_mean = nullptr, *ring_sd = nullptr;
    int ring_K = 0, ring_next = 0;
    int64_t ring_count = 0;

    // profiling of the sweeps
    bool prof = false;
    int prof_stride = 1;
    int64_t prof_seen = 0;
    std::vector<hipEvent_t> ev;
    size_t ev_used = 0;
    double prof_ms_acc = 0.0;
    int64_t prof_launches = 0;
};

static int fail(gh_ctx *c, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_create_error = buf;
    return code;
}

#define HIPCHK(c, call)                                                                     \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            return fail((c), e_ == hipErrorOutOfMemory ? GH_ERR_NOMEM : GH_ERR_HIP,         \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__,    \
                        __LINE__);                                                          \
    } while (0)

template <typename T>
static int dalloc(gh_ctx *c, T **out, size_t count, bool zero = true)
{
    if (*out) return GH_OK;
    void *p = nullptr;
    size_t bytes = (count ? count : 1) * sizeof(T);
    HIPCHK(c, hipMalloc(&p, bytes));
    c->allocs.push_back(p);
    if (zero) HIPCHK(c, hipMemsetAsync(p, 0, bytes, c->stream));
    *out = static_cast<T *>(p);
    return GH_OK;
}

#define TRY(x)                 \
    do {                       \
        int rc_ = (x);         \
        if (rc_ != GH_OK) return rc_; \
    } while (0)

static int h2d(gh_ctx *c, double *dst, const double *src, size_t n)
{
    HIPCHK(c, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));  // caller-owned pageable memory: do not outlive the call
    return GH_OK;
}

static int d2h(gh_ctx *c, double *dst, const double *src, size_t n)
{
    HIPCHK(c, hipMemcpyAsync(dst, src, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

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
    return pick_sweep<16>(c->EPT2, c->PF, c->NT);
}

static weight_fn weight_for(const gh_ctx *c)
{
    if (c->TW == 1) return pick_weight<1>(c->EPT2);
    if (c->TW == 4) return pick_weight<4>(c->EPT2);
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
    else if (ld <= 16384) tw = 16;
    else {
        // more rows than a team can hold in registers: row panels of <= 16384 rows.  The dot
        // product of a column then spans several launches, so the adjoint and the forward can no
        // longer share one read of G (two reads per step, like the reference's formulation).
        tw = 16;
        c->n_panels = (int)((ld + 10239) / 10240);  // <= 10240 rows: 5 double2 per thread, no spills
        c->panel_rows = ((ld + c->n_panels - 1) / c->n_panels + 15) / 16 * 16;
    }
    per = tw * 128;
    int e = (int)((c->panel_rows + per - 1) / per);
    if (e == 7) e = 8;
    c->TW = tw;
    c->EPT2 = e;
    // two columns in flight per team where the registers allow it (16-wave teams with <= 5 double2
    // per thread: 122 VGPRs, no spills; measured at 6 and 8 double2: 5.4 / 3.1 TB/s against 6.5 / 6.1
    // with one column in flight)
    c->PF = env_int("GRAVHMC_PF", (tw == 16 && e <= 5) ? 2 : 1) == 2 ? 2 : 1;
    // G larger than the Infinity Cache is streamed once per sweep: bypass-friendly loads
    c->NT = env_int("GRAVHMC_NT", c->ld * c->M * 8 > (int64_t)(512 << 20) ? 1 : 0) != 0;
    const int wg_teams = (tw == 1) ? 4 : 1;
    // resident workgroups per CU we size the grid for (register/LDS budget of the kernel)
    int wg_per_cu = (tw == 16) ? 1 : 4;
    wg_per_cu = env_int("GRAVHMC_WG_PER_CU", wg_per_cu);
    int64_t max_teams = (int64_t)c->cus * wg_per_cu * wg_teams;
    int64_t min_cols = env_int("GRAVHMC_MIN_COLS", tw == 1 ? 2 : 1);
    int64_t cpt = (c->M + max_teams - 1) / max_teams;
    if (cpt < min_cols) cpt = min_cols;
    c->cols_per_team = cpt;
    c->n_teams = (int)((c->M + cpt - 1) / cpt);
    c->n_teams_sweep = c->n_teams;
    c->grid = (c->n_teams + wg_teams - 1) / wg_teams;
    if (c->n_panels > 1) c->n_teams = std::max(c->n_teams, (int)((c->M + 255) / 256));  // vec_update partials
    c->lds_bytes = (size_t)(tw == 1 ? 5 * ld : c->panel_rows + 2 * (tw + 8)) * sizeof(double);
    if (c->lds_bytes > 160 * 1024) return fail(c, GH_ERR_UNSUPPORTED, "LDS budget exceeded");
    sweep_fn f = sweep_for(c);
    if (!f) return fail(c, GH_ERR_UNSUPPORTED, "no sweep instantiation for EPT2=%d", e);
    HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(f), c->lds_bytes));
    return GH_OK;
}

static MfGeom mf_geom(const gh_ctx *c)
{
    MfGeom g;
    g.kind = c->cell_kind;
    g.N = c->N;
    g.M = c->M;
    if (c->cell_kind == GH_CELL_TESSEROID) {
        g.o0 = c->tconv;
        g.o1 = c->tconv + c->N;
        g.o2 = c->tconv + 2 * c->N;
        g.o3 = c->tconv + 3 * c->N;
    } else {
        g.o0 = c->obs[0];
        g.o1 = c->obs[1];
        g.o2 = c->obs[2];
        g.o3 = nullptr;
    }
    g.bounds6 = c->bounds;
    g.ratio = c->ratio;
    return g;
}

// matrix-free counterpart of one sweep: adjoint/update pass, then forward pass
static int launch_mf(gh_ctx *c, SweepArgs &a)
{
    const MfGeom g = mf_geom(c);
    const double *wm = c->weighted ? c->wm : nullptr;
    bool timed = c->prof && c->ev_used + 2 <= c->ev.size();
    if (timed) HIPCHK(c, hipEventRecord(c->ev[c->ev_used], c->stream));
    if (a.mode & SW_ADJ) {
        if (!wm) return fail(c, GH_ERR_ARG, "matrix-free adjoint needs gh_weight first");
        mf_adjoint_kernel<<<dim3((unsigned)((c->M + 3) / 4)), dim3(256), 0, c->stream>>>(g, a, wm);
    }
    if (a.mode & SW_FWD) {
        const double *x = (a.mode & SW_UPD) ? a.x_out : a.x_in;
        mf_forward_kernel<<<dim3((unsigned)((c->ld + 255) / 256), (unsigned)c->grid), dim3(256), 0,
                            c->stream>>>(g, x, wm, c->mf_cells_per_chunk, c->ld, a.slab);
    }
    if (timed) {
        HIPCHK(c, hipEventRecord(c->ev[c->ev_used + 1], c->stream));
        c->ev_used += 2;
    }
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
        c->ev_used += 2;
    }
    if (c->prof) c->prof_launches += 1;
    HIPCHK(c, hipGetLastError());
    return GH_OK;
}

static int launch_sweep(gh_ctx *c, SweepArgs &a)
{
    if (c->mf) return launch_mf(c, a);
    if (c->n_panels == 1) {
        a.row0 = 0;
        a.rows = c->ld;
        return launch_sweep_one(c, a);
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
            vec_update_kernel<<<dim3((unsigned)((c->M + 255) / 256)), dim3(256), 0, c->stream>>>(u, gdst, c->M);
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

// ------------------------------------------------------------------ collective layer

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t,
                              hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

// RCCL is bound at run time: the copy already in the process (e.g. the one torch.distributed
// loaded) wins, else the ROCm installation's.
static RcclApi *rccl_api(std::string &err)
{
    static RcclApi api;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (api.handle) {
            api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
            api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
            api.AllReduce = (decltype(api.AllReduce))dlsym(api.handle, "ncclAllReduce");
            api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
            api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
        }
    }
    if (!api.handle || !api.GetUniqueId || !api.CommInitRank || !api.AllReduce) {
        err = "RCCL (librccl.so) could not be loaded";
        return nullptr;
    }
    return &api;
}

// In-place sum over ranks of `count` doubles at device pointer `buf`, ordered on the stream.
static int comm_allreduce(gh_ctx *c, double *buf, int64_t count)
{
    gh_ctx::Shard &sh = c->sh;
    if (sh.kind == 0) return GH_OK;
    sh.collectives += 1;
    if (sh.kind == 1) {
        std::string err;
        RcclApi *api = rccl_api(err);
        if (!api) return fail(c, GH_ERR_COMM, "%s", err.c_str());
        ncclResult_t r = api->AllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, sh.comm, c->stream);
        if (r != ncclSuccess)
            return fail(c, GH_ERR_COMM, "ncclAllReduce: %s", api->GetErrorString ? api->GetErrorString(r) : "error");
        return GH_OK;
    }
    // host-staged reducer (e.g. gloo): device -> pinned host -> callback -> device
    if ((size_t)count > sh.buf_n) return fail(c, GH_ERR_ARG, "all-reduce larger than the staging buffer");
    HIPCHK(c, hipMemcpyAsync(sh.hbuf, buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (sh.cb(sh.user, sh.hbuf, count) != 0) return fail(c, GH_ERR_COMM, "all-reduce callback failed");
    HIPCHK(c, hipMemcpyAsync(buf, sh.hbuf, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    return GH_OK;
}

// The same for a few host scalars (count <= ld).
static int comm_allreduce_host(gh_ctx *c, double *hv, int64_t count)
{
    gh_ctx::Shard &sh = c->sh;
    if (sh.kind == 0) return GH_OK;
    if (count > c->ld) return fail(c, GH_ERR_ARG, "gh_shard_allreduce: count too large");
    if (sh.kind == 2) {
        sh.collectives += 1;
        if (sh.cb(sh.user, hv, count) != 0) return fail(c, GH_ERR_COMM, "all-reduce callback failed");
        return GH_OK;
    }
    HIPCHK(c, hipMemcpyAsync(sh.buf, hv, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    TRY(comm_allreduce(c, sh.buf, count));
    HIPCHK(c, hipMemcpyAsync(hv, sh.buf, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return GH_OK;
}

static int shard_common_init(gh_ctx *c, int rank, int world, int64_t M_global, int64_t m0)
{
    if (world < 1 || rank < 0 || rank >= world) return fail(c, GH_ERR_ARG, "gh_shard_init: bad rank/world");
    if (m0 < 0 || m0 + c->M > M_global) return fail(c, GH_ERR_ARG, "gh_shard_init: cell range outside the model");
    if (c->wv.on) return fail(c, GH_ERR_UNSUPPORTED, "sharding with the wavelet forward is not supported");
    c->sh.rank = rank;
    c->sh.world = world;
    c->sh.M_global = M_global;
    c->sh.m0 = m0;
    TRY(dalloc(c, &c->sh.buf, (size_t)c->ld + 8));
    if (!c->sh.hbuf) HIPCHK(c, hipHostMalloc((void **)&c->sh.hbuf, sizeof(double) * ((size_t)c->ld + 8)));
    c->sh.buf_n = (size_t)c->ld + 8;
    c->chain_ready = false;
    return GH_OK;
}

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
    HIPCHK(c, hipMemsetAsync(w.coeff, 0, sizeof(double) * (size_t)w.Mp, c->stream));
    TRY(run_dwt(c, x, c->M, 1, w.coeff, w.s1, w.s2));
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

// slab -> d ; regulariser ; residual + scalars.  x: position the forward belongs to.
// slab (c->grid rows) -> d_out (+ per-block partial sums of d + grav_fix).  Many slab rows
// (small problems spread over many workgroups) are summed in two passes so that no thread walks
// hundreds of rows serially.
static void reduce_slab(gh_ctx *c, const double *gfix, double *d_out)
{
    const int rows = c->grid;
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
    } else if (c->grid > 64 && c->slab2) {
        // many slab rows: first stage of the reduction and the regulariser share one launch,
        // finish_kernel sums the 16 segments
        nseg = c->slab2_rows;
        reduce_reg_kernel<<<dim3((unsigned)(c->n_dpart * nseg + c->n_regpart)), dim3(256), 0, c->stream>>>(
            c->slab, c->grid, c->ld, nseg, c->n_dpart, c->slab2, ra);
        src = c->slab2;
    } else {
        reg_kernel<<<dim3(c->n_regpart), dim3(256), 0, c->stream>>>(ra);
        src = c->slab;
        nseg = c->grid;
    }
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
    if (c->n_panels > 1) TRY(dalloc(c, &c->gbuf, M));
    TRY(dalloc(c, &c->slab, (size_t)c->grid * ld));
    if (c->grid > 64) {
        c->slab2_rows = 16;
        TRY(dalloc(c, &c->slab2, (size_t)c->slab2_rows * ld));
    }
    c->n_dpart = (int)((c->ld + 31) / 32);
    c->n_regpart = (int)((c->M + 255) / 256);
    c->n_pp0 = (int)std::min<int64_t>(1024, (c->M + 255) / 256);
    TRY(dalloc(c, &c->dpart, (size_t)c->n_dpart));
    TRY(dalloc(c, &c->regpart, (size_t)c->n_regpart));
    TRY(dalloc(c, &c->pp_part, (size_t)c->n_teams));
    TRY(dalloc(c, &c->ppn_part, (size_t)c->n_teams));
    TRY(dalloc(c, &c->pp0_part, (size_t)c->n_pp0));
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
        c->h_scal_n = 16 + 2 * (size_t)c->n_teams + (size_t)c->n_pp0;
        HIPCHK(c, hipHostMalloc((void **)&c->h_scal, c->h_scal_n * sizeof(double)));
    }
    return GH_OK;
}

static int need(gh_ctx *c, bool cond, const char *what)
{
    if (!cond) return fail(c, GH_ERR_ARG, "%s", what);
    return GH_OK;
}

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
    hipSetDevice(c->device);
    if (c->stream) hipStreamSynchronize(c->stream);
    if (c->sh.comm) {
        std::string err;
        RcclApi *api = rccl_api(err);
        if (api && api->CommDestroy) api->CommDestroy(c->sh.comm);
    }
    if (c->sh.hbuf) hipHostFree(c->sh.hbuf);
    if (c->bt.h) hipHostFree(c->bt.h);
    for (void *p : c->allocs) hipFree(p);
    if (c->h_scal) hipHostFree(c->h_scal);
    for (hipEvent_t ev : c->ev) hipEventDestroy(ev);
    if (c->rs.ev0) hipEventDestroy(c->rs.ev0);
    if (c->rs.ev1) hipEventDestroy(c->rs.ev1);
    if (c->stream) hipStreamDestroy(c->stream);
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
    c->mf = enable != 0;
    if (c->mf) {
        // partition used by the matrix-free passes: one wave per cell (adjoint), chunks of
        // cells per forward partial
        c->n_teams = (int)((c->M + 3) / 4);
        const int64_t chunks = std::min<int64_t>(c->M, std::max<int64_t>(1, (int64_t)c->cus * 16 / std::max<int64_t>(1, (c->ld + 255) / 256)));
        c->mf_cells_per_chunk = (c->M + chunks - 1) / chunks;
        c->grid = (int)((c->M + c->mf_cells_per_chunk - 1) / c->mf_cells_per_chunk);
    }
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
        if (c->cell_kind == GH_CELL_TESSEROID) {
            const int64_t N = c->N;
            TRY(dalloc(c, &c->tconv, (size_t)(4 * N)));
            tess_convert_kernel<<<dim3((unsigned)((N + 255) / 256)), dim3(256), 0, c->stream>>>(
                c->obs[0], c->obs[1], c->obs[2], N, c->tconv, c->tconv + N, c->tconv + 2 * N,
                c->tconv + 3 * N);
            HIPCHK(c, hipGetLastError());
        }
        c->have_G = true;
        c->weighted = false;
        c->chain_ready = false;
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
    if (c->mf) {
        mf_colnorm_kernel<<<dim3((unsigned)((c->M + 3) / 4)), dim3(256), 0, c->stream>>>(mf_geom(c), weightfactor,
                                                                                        c->wm);
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
    const double mean = PW::sum(t.data(), N) / (double)N;
    for (auto &v : t) v -= mean;
    TRY(h2d(c, c->dobs_c, t.data(), N));
    c->have_fix = grav_fix != nullptr;
    if (grav_fix) TRY(h2d(c, c->gfix, grav_fix, N));
    c->have_data = true;
    c->chain_ready = false;
    return GH_OK;
}

int gh_set_reg(gh_ctx *c, int kind, double alpha, double beta, const int shape3[3], const double *mwapr)
{
    if (!c || !mwapr) return fail(c, GH_ERR_ARG, "gh_set_reg: null pointer");
    if (kind < 0 || kind > 3)
        return fail(c, GH_ERR_ARG, "Please choose regularization from 'MS','Damping', 'Smoothness', 'TV'.");
    const bool stencil = (kind == GH_REG_SMOOTHNESS || kind == GH_REG_TV);
    if (c->sh.kind != 0 && stencil) {
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
    if (c->sh.kind != 0 && stencil) {
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
    TRY(comm_allreduce(c, c->tmpN, c->ld));
    return d2h(c, dpre, c->tmpN, (size_t)c->N);
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
    HIPCHK(c, hipMemcpyAsync(c->h_scal, o.scal, 4 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TRY(d2h(c, grad, c->tmpM, (size_t)c->M));
    if (dpre) TRY(d2h(c, dpre, o.d, (size_t)c->N));
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
    if (c->sh.kind != 0) return fail(c, GH_ERR_UNSUPPORTED, "wavelet forward on a sharded kernel is not supported");
    if (c->mf) return fail(c, GH_ERR_UNSUPPORTED, "wavelet compression needs the stored kernel (not matrix-free)");
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
            gather_rows_kernel<<<dim3((unsigned)((M + 31) / 32), (unsigned)((nr + 31) / 32)), dim3(256), 0,
                                 c->stream>>>(c->G, c->ld, M, i0, nr, X);
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
    TRY(dalloc(c, &w.coeff, (size_t)Mp));
    TRY(dalloc(c, &w.s1, (size_t)Mp));
    TRY(dalloc(c, &w.s2, (size_t)Mp));
    w.on = true;
    w.F_valid = false;
    c->rs.state = 0;  // plan the resident chain kernel again (it would need the dense form)
    c->chain_ready = false;
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
    c->accept_count = 0;
    TRY(h2d(c, c->xb[0], x0, (size_t)c->M));
    TRY(h2d(c, c->low, low, (size_t)c->M));
    TRY(h2d(c, c->high, high, (size_t)c->M));
    TRY(eval_forward(c, c->xb[0], c->st[0]));
    TRY(d2h(c, c->h_scal, c->st[0].scal, 4));
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

static inline int other_of3(int a, int b)
{
    for (int i = 0; i < 3; ++i)
        if (i != a && i != b) return i;
    return 0;
}

int gh_chain_trajectory(gh_ctx *c, const double *p0, double dt, int L, double u, int *accepted,
                        double out5[5])
{
    if (!c || !p0 || !accepted || !out5) return fail(c, GH_ERR_ARG, "gh_chain_trajectory: null pointer");
    TRY(need(c, c->chain_ready, "gh_chain_trajectory: call gh_chain_init first"));
    if (L < 1) return fail(c, GH_ERR_ARG, "gh_chain_trajectory: L must be >= 1");
    HIPCHK(c, hipSetDevice(c->device));
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
    const bool spec = c->pn_valid;
    const double probe[3] = {c->pn_probe[0], c->pn_probe[1], c->pn_probe[2]};
    const double pn_pp0 = c->pn_pp0;
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
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double pp1 = 0.0, pp0 = 0.0;
    for (int t = 0; t < nt; ++t) pp1 += h[16 + t];
    if (use_spec)
        pp0 = c->spec_pp0;
    else
        for (int t = 0; t < c->n_pp0; ++t) pp0 += h[16 + 2 * nt + t];
    double pn_pp0_g = pn_pp0;
    if (c->sh.kind != 0) {
        // kinetic energies are sums over cells: combine the ranks' parts (same bits everywhere)
        double v[3] = {pp1, use_spec ? 0.0 : pp0, spec ? pn_pp0 : 0.0};
        TRY(comm_allreduce_host(c, v, 3));
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

typedef void (*resident_fn)(ResArgs);
enum { GH_RESIDENT_ABORTED = 1000 };  // internal: chain_run_resident gave up, state untouched

// rc: double2 chunks per lane and column; cw: columns a wave keeps in registers (0: none, dots
// read LDS).  Only register copies of at most 20 double2 (80 VGPRs: no spills) are compiled.
extern "C++" {
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
}  // extern "C++"

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
    const int cpw = (int)((c->M + c->cus - 1) / c->cus);
    const size_t lds = resident_lds_doubles(c->ld, cpw, 1) * sizeof(double);
    if (lds > (size_t)lds_max || cpw > RES_THREADS) return false;
    r.lds_max = lds_max;
    r.cpw = cpw;
    r.nwg = (int)((c->M + cpw - 1) / cpw);
    if (r.nwg > RES_MAX_WG) return false;
    r.rc = (int)((c->ld / 2 + 63) / 64);
    // columns per wave for the register copy of the dots pass (0: the wave has more than 4)
    r.ct = (env_int("GRAVHMC_RESIDENT_REGS", 1) && cpw <= 4 * RES_WAVES) ? (cpw + RES_WAVES - 1) / RES_WAVES : 0;
    if (r.ct * ((int)((c->ld / 2 + 63) / 64)) > 20) r.ct = 0;  // (what resident_for compiles)
    // wavelet-compressed forward: LDS holds its dense model-space form, the dots need their own
    // (register) copy of Aw
    if (c->wv.on && (r.ct == 0 || wavelet_dense_form(c) != GH_OK)) return false;
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
    const size_t lds = resident_lds_doubles(c->ld, r.cpw, q.C) * sizeof(double);
    if (q.C < 1 || q.C > RES_MAX_CHAINS || lds > (size_t)r.lds_max)
        return fail(c, GH_ERR_ARG, "resident chain kernel: %d chains do not fit the LDS", q.C);
    HIPCHK(c, allow_dynamic_lds(reinterpret_cast<const void *>(resident_for(r.rc, r.ct)), lds));
    if (!r.slabg) {
        // (+8 rows / entries: the abort test announces one phantom workgroup per cluster)
        TRY(dalloc(c, &r.slabg, (size_t)(r.nwg + 8) * (size_t)c->ld * 2));
        TRY(dalloc(c, &r.xslabg, 2 * (size_t)RES_CLUSTERS * (size_t)c->ld * 2));
        TRY(dalloc(c, &r.dclg, (size_t)RES_CLUSTERS * (size_t)c->ld * 2));
        TRY(dalloc(c, &r.scalg, (size_t)r.nwg * 8));
        TRY(dalloc(c, &r.xccg, (size_t)r.nwg + 8));
        TRY(dalloc(c, &r.xpub, 2 * M));
        TRY(dalloc(c, &r.abort_w, 4));
        TRY(dalloc(c, &r.n_run, 4));
        if (env_int("GRAVHMC_RESIDENT_TIMING", 0)) TRY(dalloc(c, &r.dbg, 32));
        HIPCHK(c, hipEventCreate(&r.ev0));
        HIPCHK(c, hipEventCreate(&r.ev1));
    }
    if (K > r.Kcap) {
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
    if ((uint64_t)r.tag + (uint64_t)steps + (uint64_t)q.C + 2 > 0xf0000000ull ||
        (uint64_t)r.tagE + (uint64_t)K + (uint64_t)q.C + 2 > 0xf0000000ull) {
        // 32-bit tags about to wrap: start the count again on zeroed granules
        HIPCHK(c, hipMemsetAsync(r.slabg, 0, (size_t)r.nwg * (size_t)c->ld * 2 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.xslabg, 0, 2 * (size_t)RES_CLUSTERS * (size_t)c->ld * 2 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.dclg, 0, (size_t)RES_CLUSTERS * (size_t)c->ld * 2 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.scalg, 0, (size_t)r.nwg * 8 * sizeof(ghk::u64), c->stream));
        HIPCHK(c, hipMemsetAsync(r.xccg, 0, (size_t)r.nwg * sizeof(ghk::u64), c->stream));
        r.tag = r.tagE = 0;
    }
    HIPCHK(c, hipMemsetAsync(r.abort_w, 0, 4 * sizeof(unsigned), c->stream));
    HIPCHK(c, hipMemcpyAsync(r.p0s, q.p0s, (size_t)K * M * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(r.us, q.us, (size_t)K * sizeof(double), hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(r.L, q.L, (size_t)K * sizeof(int), hipMemcpyHostToDevice, c->stream));
    if (q.chain_of)
        HIPCHK(c, hipMemcpyAsync(r.chain, q.chain_of, (size_t)K * sizeof(int), hipMemcpyHostToDevice, c->stream));
    ResArgs a{};
    a.G = c->G;
    a.Gl = c->wv.on ? c->wv.F : c->G;
    a.ld = c->ld;
    a.N = c->N;
    a.M = c->M;
    a.cols_per_wg = r.cpw;
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
    HIPCHK(c, hipMemcpyAsync(accepted, r.accepted, (size_t)K * sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(out5s, r.out5s, (size_t)K * 5 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (h_sync[0] != 0u) {
        // A workgroup waited 2 s for the others: they were not all resident (another process holding
        // compute units of this device).  Nothing of the chain state was written; this context goes
        // back to the sweep-per-launch path for good and the caller's batch is run there.
        r.state = -1;
        fprintf(stderr, "libgravhmc: resident chain kernel timed out waiting for its workgroups; "
                        "continuing on the sweep-per-launch path\n");
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
    // bring the per-launch state (d, r, scalars of the current sample) back in step with x
    c->spec_valid = c->pn_valid = false;
    TRY(eval_forward(c, c->xb[c->xcur], c->st[c->cur]));
    TRY(d2h(c, c->h_scal, c->st[c->cur].scal, 4));
    c->U_cur[0] = c->h_scal[2];
    c->U_cur[1] = c->h_scal[0];
    c->U_cur[2] = c->h_scal[1];
    return GH_OK;
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
    if (resident_plan(c)) {
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
        if (nxt) TRY(gh_chain_prefetch_momentum(c, nxt));
        TRY(gh_chain_trajectory(c, p0s + (size_t)k * M, dt, L[k], us[k], &accepted[k], out5s + 5 * k));
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
    return d2h(c, dsyn, c->st[c->cur].d, (size_t)c->N);
}

int gh_chain_stats(gh_ctx *c, int64_t *spec_hits, int64_t *spec_misses)
{
    if (!c) return GH_ERR_ARG;
    if (spec_hits) *spec_hits = c->spec_hits;
    if (spec_misses) *spec_misses = c->spec_misses;
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
    HIPCHK(c, hipHostMalloc((void **)&b.h, sizeof(double) * (size_t)(CB * 4 + (b.n_waves + b.n_pp0) * CB)));
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
        c->ev_used += 2;
    }
    c->bt.sweeps += 1;
    return GH_OK;
}

// forward of all chains at X, then regulariser and residuals into (D, GREG, Rt, scal)
static int batch_evaluate(gh_ctx *c, const double *X, double *D, double *GREG, double *Rt)
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
    fa.scal = b.scal;
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

int gh_batch_init(gh_ctx *c, int C, const double *x0s, const double *low, const double *high)
{
    if (!c || !x0s || !low || !high) return fail(c, GH_ERR_ARG, "gh_batch_init: null pointer");
    if (C < 1 || C > CB) return fail(c, GH_ERR_ARG, "gh_batch_init: 1..16 chains per batch");
    TRY(need(c, c->have_G && c->have_data && c->have_reg && !c->mf,
             "gh_batch_init: needs the stored (dense) kernel matrix, gh_set_data and gh_set_reg"));
    if (c->wv.on || c->sh.kind != 0)
        return fail(c, GH_ERR_UNSUPPORTED, "batched chains run on the dense, unsharded kernel only");
    HIPCHK(c, hipSetDevice(c->device));
    TRY(ensure_work(c));
    TRY(h2d(c, c->low, low, (size_t)c->M));
    TRY(h2d(c, c->high, high, (size_t)c->M));
    gh_ctx::Resident &r = c->rs;
    r.b_on = false;
    if (resident_plan(c) &&
        resident_lds_doubles(c->ld, r.cpw, C) * sizeof(double) <= (size_t)r.lds_max) {
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
    HIPCHK(c, hipSetDevice(c->device));
    const int C = b.C;
    int Lmax = 0;
    for (int k = 0; k < C; ++k) {
        if (L[k] < 1) return fail(c, GH_ERR_ARG, "gh_batch_trajectory: L must be >= 1");
        Lmax = std::max(Lmax, L[k]);
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
        bool timed;
        TRY(batch_time_begin(c, timed));
        batch_adjoint_kernel<<<dim3((unsigned)(b.n_waves / 4)), dim3(256), 0, c->stream>>>(a);
        TRY(batch_time_end(c, timed));
        HIPCHK(c, hipGetLastError());
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

int gh_batch_get_x(gh_ctx *c, int chain, double *x)
{
    if (!c || !x) return fail(c, GH_ERR_ARG, "gh_batch_get_x: null pointer");
    TRY(need(c, c->bt.ready && chain >= 0 && chain < c->bt.C, "gh_batch_get_x: no such chain"));
    HIPCHK(c, hipSetDevice(c->device));
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
    if (!c->rs.dbg) return GH_OK;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipMemcpy(out32, c->rs.dbg, 32 * sizeof(long long), hipMemcpyDeviceToHost));
    return GH_OK;
}

int gh_debug_stream_read(gh_ctx *c, int blocks, int threads, int nt, int reps, double *ms_out);

int gh_measure_stream_read(gh_ctx *c, int nt, int reps, double *ms_per_pass)
{
    if (!c || !ms_per_pass || reps < 1) return fail(c, GH_ERR_ARG, "gh_measure_stream_read: bad arguments");
    TRY(need(c, c->have_G && !c->mf, "gh_measure_stream_read: needs a stored kernel matrix"));
    return gh_debug_stream_read(c, c->cus * 2, 1024, nt, reps, ms_per_pass);
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

int gh_shard_allreduce(gh_ctx *c, double *host_buf, int64_t count)
{
    if (!c || !host_buf) return fail(c, GH_ERR_ARG, "gh_shard_allreduce: null pointer");
    HIPCHK(c, hipSetDevice(c->device));
    return comm_allreduce_host(c, host_buf, count);
}

int gh_profile_enable(gh_ctx *c, int enable)
{
    if (!c) return GH_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    if (enable && c->ev.empty()) {
        c->ev.resize(8192);
        for (auto &e : c->ev) HIPCHK(c, hipEventCreate(&e));
    }
    c->prof = enable != 0;
    c->prof_stride = (c->ld * c->M * 8 < (int64_t)(1 << 30)) ? 16 : 1;
    c->prof_seen = 0;
    c->ev_used = 0;
    c->prof_ms_acc = 0.0;
    c->prof_launches = 0;
    c->prof_res_evals = 0;
    return GH_OK;
}

int gh_profile_read(gh_ctx *c, double *sweep_ms, int64_t *sweep_launches, int64_t *bytes_per_sweep)
{
    if (!c) return GH_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    double ms = c->prof_ms_acc;
    for (size_t i = 0; i + 1 < c->ev_used; i += 2) {
        float t = 0.f;
        HIPCHK(c, hipEventElapsedTime(&t, c->ev[i], c->ev[i + 1]));
        ms += t;
    }
    // launches beyond the event pool are counted but not timed: scale to the timed share
    // (an evaluation inside the resident chain kernel counts as one sweep)
    const int64_t timed = (int64_t)(c->ev_used / 2) + c->prof_res_evals;
    if (sweep_ms) *sweep_ms = ms;
    if (sweep_launches) *sweep_launches = timed;
    // one launch reads one row panel of G (the whole matrix when N <= 16384)
    if (bytes_per_sweep)
        *bytes_per_sweep = (c->n_panels > 1 ? c->panel_rows : c->N) * c->M * (int64_t)sizeof(double);
    return GH_OK;
}

}  // extern "C"
