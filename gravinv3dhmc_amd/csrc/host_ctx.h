// libgravhmc host side: the context (one GPU + one stream + one inversion problem resident in
// HBM) and the small helpers every other part uses.  Included once by gravhmc.hip.
#pragma once

static thread_local std::string g_create_error;

struct LonSymHost;  // host_lonsym.h

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct gh_ctx {
    int device = 0;
    int64_t N = 0, M = 0, ld = 0;
    int cus = 0;
    hipStream_t stream = nullptr;
    hipStream_t copy_stream = nullptr;  // upload of the next trajectory's momentum beside the sweeps
    hipEvent_t copy_ev = nullptr;
    std::string err;
    std::vector<void *> allocs;

    // geometry / kernel
    double *obs[3] = {nullptr, nullptr, nullptr};
    double *bounds = nullptr;
    bool obs_h_uniform = false;  // every observation at the same height (third coordinate)
    double obs_h0 = 0.0;
    int cell_kind = -1;
    double ratio = 1.6;
    bool have_obs = false, have_cells = false, have_G = false, weighted = false;
    double *G = nullptr;
    int64_t warn_cells = 0, leaves = 0;
    bool mf = false;          // matrix-free: entries are re-evaluated, G is never stored
    bool mf_before_ls = false;  // what gh_set_matrix_free asked for, while the shift-invariant store holds mf
    bool dense_ok = true;     // N fits the register-resident sweep (<= 16384 rows)
    double *tconv = nullptr;  // tesseroid obs converted to (lon rad, sin lat, cos lat, radius)
    int64_t mf_cells_per_chunk = 0;
    // fused matrix-free pass (mf_fused_kernel: every entry evaluated once per step), N <= 16384
    bool mf_fused = false;
    int mf_T = 0, mf_EPT = 0;
    size_t mf_lds = 0;
    double *mf_cellc = nullptr;       // tesseroids: per-cell constants (tess_cellconst_kernel)
    ghk::MfStats *mf_stats = nullptr; // entries / GLQ leaves evaluated while profiling is enabled
    // tesseroids: near-field table (pairs that need the adaptive subdivision: evaluated once, kept)
    bool mf_near_on = false;
    bool mf_pipe = false;    // mf_tess_fast_kernel (next column's constants fetched ahead; r and the column in LDS)
    bool mf_exact = false;   // the root leaf in the reference's operation order (tess_leaf_cc) instead of tess_leaf_fast
    int mf_exact_req = -1;   // gh_set_matrix_free_exact: 0 / 1; -1: not called, the environment (GRAVHMC_MF_EXACT) decides
    int64_t *mf_near_ptr = nullptr;
    int *mf_near_row = nullptr;
    double *mf_near_val = nullptr;
    int64_t mf_near_n = 0, mf_near_leaves = 0;
    int64_t mf_launches = 0;
    // single chain on teams of workgroups (mf_team_kernel, mfbatch.hip.h)
    struct MfTeam {
        int state = 0;  // 0 not planned, 1 usable, -1 not applicable / given up
        int members = 0, ranges = 0, tpr = 0;
        ghk::u64 *gran = nullptr;
        unsigned *abort_w = nullptr;
        unsigned tag = 0;
        double *snear = nullptr;
        bool inflight = false;
        int aborts = 0;
        int64_t launches = 0;
    } mft;
    bool chain_teams_ok = false;  // set by the trajectory code around its sweeps (it handles a time-out)
    // shift-invariant store of a regular spherical grid (lonsym.hip.h): a table instead of G or of
    // per-step evaluations; a flavour of the matrix-free mode (gh_set_shift_invariant)
    LonSymHost *ls = nullptr;

    // chains of a batch on the shift-invariant store: light contexts of their own (stream, chain state, work
    // buffers) that share this context's tables and problem vectors; run concurrently, one host thread each
    std::vector<gh_ctx *> kids;

    // sweep configuration
    int TW = 0, EPT2 = 0, PF = 1;
    int n_panels = 1;        // row panels of the dense sweep (N > 16384 rows: > 1, two reads of G per step)
    int64_t panel_rows = 0;
    double *gbuf = nullptr;   // gradient accumulated over the panels
    // team sweep (teamsweep.hip.h): N > 16384 with ONE read of G per fused step
    struct Team {
        int state = 0;        // 0 not planned, 1 usable, -1 not applicable / given up
        int Q = 0, tpx = 0, grid = 0;
        int threads = 0, ept2 = 0, depth = 0;  // instantiation of teamsweep_kernel
        int lag = 1;                           // columns between a member's dot and its update
        int64_t panel_rows = 0;                // rows per member
        int64_t cols_per_team = 0;
        size_t lds = 0;
        ghk::u64 *gran = nullptr;
        unsigned *abort_w = nullptr;
        unsigned tag = 0;
        bool inflight = false;  // launched since the abort word was last looked at
        int aborts = 0;
        int64_t launches = 0;
        unsigned late_polls = 0;
    } tm;
    int slab_live = 0;        // rows of the slab the last forward launch wrote
    // one-launch epilogue (reduce_finish_kernel): sums of the slab rows from the sweep, |r|^2 partials
    double *dsum = nullptr;
    bool dsum_live = false;   // the last forward launch delivered dsum for all slab_live rows
    int dsum_n = 0;           // > 0: dsum holds this many partial sums (not one per slab row)
    double gfix_sum = 0.0;
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
        double *part = nullptr;        // |r|^2 and R partials of a one-launch epilogue
        mutable bool pending = false;  // scal[0..2] still to be summed from `part` (scal_ready)
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
           *ppn_part = nullptr, *pp0_part = nullptr, *pn0_part = nullptr, *scal_all = nullptr;
    double *tmpM = nullptr, *tmpN = nullptr;
    int n_dpart = 0, n_regpart = 0, n_pp0 = 0;
    double *h_scal = nullptr;  // pinned: scalars + partial sums
    size_t h_scal_n = 0;
    bool chain_ready = false;
    double U_cur[3] = {0, 0, 0};
    bool st_stale = false;  // the resident chain kernel moved x: st[cur] / U_cur are those of an older sample

    // column-block sharding of ONE chain over several GPUs: this context holds
    // the cells [m0, m0 + M) of M_global; N-vectors are replicated, the forward partials are
    // summed across ranks once per potential evaluation.
    struct Shard {
        int kind = 0;  // 0 single GPU, 1 RCCL all-reduce on the stream, 2 host callback
        int axis = 0;  // 0: the CELLS (columns of G) are split over the ranks; 1: the OBSERVATIONS (rows of G)
        int64_t N_global = 0, n0 = 0;  // axis 1: observations of the whole problem, first one of this rank
        double *rbuf = nullptr;        // axis 1: {sum of d + grav_fix, |r|^2} on their way through the all-reduces
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
        int64_t coeff_n = 0;                                    // doubles each of them holds
        ghk::DwtLdsPlan lds_plan;  // one-launch transform of one model vector (dwt_lds_kernel)
        size_t lds_bytes = 0;      // 0: the working block does not fit the LDS -> one launch per pass
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
        double *stage2 = nullptr; // second set of rows (gh_batch_run: next trajectories' momenta sent ahead)
        // gh_batch_run: second working set (the sweep reads one, the evaluation writes the other, so the
        // proposal's values survive a speculative first step) and the next trajectories' momenta
        double *GREGw2 = nullptr, *Dw2 = nullptr, *Rtw2 = nullptr, *scal2 = nullptr, *Pn = nullptr, *pn0_part = nullptr;
        double *Gb = nullptr;     // second copy of G in MFMA operand order (adjoint), if HBM allows
        // matrix-free batch (mfbatch.hip.h): 1 / wm, the near-field pairs as differences to the staged
        // root leaf (column-major: mf_near_ptr / mf_near_row / ndelta; row-major copy: rptr / rcol / rdelta)
        double *iw = nullptr, *Snear = nullptr, *ndelta = nullptr, *rdelta = nullptr;
        int64_t *rptr = nullptr;
        int *rcol = nullptr;
        bool mfb_near = false, near_built = false;
        int mfb_grid_adj = 0, mfb_rchunks = 0, mfb_ranges = 0, mfb_tpr = 0;
        int slab_live = 0;        // blocks of the slab the last matrix-free forward wrote
        // team pass: one evaluation (mfb_fused_kernel) / one read (batch_team_kernel, stored G) per entry and step
        bool fus_on = false, fus_inflight = false;
        int fus_members = 0, fus_ranges = 0, fus_tpr = 0, fus_aborts = 0;
        ghk::u64 *fus_gran = nullptr;
        ghk::u64 *fus_granx = nullptr;  // stored kernel on teams (batch_team_kernel): the new positions' granules
        int fus_nval = 0;               // ... and the (cell, chain) pairs of a tile a member updates
        unsigned *fus_abort = nullptr;
        long long *fus_dbg = nullptr;  // GRAVHMC_MFB_TIMING: per-phase clocks of one workgroup
        double *Pstart = nullptr;      // M x 16: the momentum each trajectory in flight started with (gh_batch_run)
        unsigned fus_tag = 0;
        int64_t fus_launches = 0;
        const double *fus_fwd_of = nullptr;  // the X whose forward partials the last fused launch left in the slab
        double *h = nullptr;      // pinned
        int n_colblocks = 0, n_regblocks = 0, n_waves = 0, n_pp0 = 0;
        int64_t cols_per_block = 0;
        double U[CB][3];
        bool ready = false;
        int64_t sweeps = 0;
        // scheduler of gh_batch_run: trajectories in flight survive the call (carry-over mode)
        struct Run {
            bool live = false;
            bool active[CB] = {};
            int s_of[CB] = {}, L_cur[CB] = {}, par[CB] = {};
            double u_cur[CB] = {}, pp0[CB] = {};
            int xi = 0, pin = 0, ws = 0;  // ws: working set the next sweep reads
            double dt = 0.0;
        } run;
    } bt;

    // resident chain kernel (resident.hip.h): G held in LDS across a whole batch of trajectories
    struct Resident {
        int state = 0;  // 0 not planned yet, 1 usable, -1 not applicable
        int cpw = 0, nwg = 0, rc = 0, ct = 0;  // ct: columns per wave kept in registers
        int lds_cols = 0;     // columns of a workgroup held in LDS
        bool split = false;   // one copy: the first 8 ct columns in registers only, the rest in LDS
        bool stream = false;  // columns beyond lds_cols are read from memory in every pass (resident.hip.h)
        size_t lds = 0;
        ghk::u64 *slabg = nullptr, *xslabg = nullptr, *dclg = nullptr, *scalg = nullptr, *xscalg = nullptr, *xccg = nullptr;
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
        int aborts = 0;               // launches that timed out (3: the context stays on the sweep path)
        bool granules_dirty = false;  // an aborted launch left tags behind: clear before the next launch
        long long *dbg = nullptr;
        hipEvent_t ev0 = nullptr, ev1 = nullptr;
        // the chains in LOCK-STEP (resbatch.hip.h): planned per gh_batch_init
        struct LockStep {
            bool on = false;
            int ks = 0, nt = 0, C = 0;
            size_t lds = 0;
            ghk::u32x4 *xslabg = nullptr, *dclg = nullptr;
            double *slabd = nullptr;
            ghk::u64 *xccg = nullptr, *flagg = nullptr;
            double *xpub = nullptr, *xs = nullptr, *ps = nullptr, *pst = nullptr, *cst = nullptr;
            double *p0s = nullptr, *us = nullptr, *out5s = nullptr, *xacc = nullptr;
            int *L = nullptr, *accepted = nullptr, *n_io = nullptr;
            double *h_stage = nullptr;       // pinned staging of the momentum rows
            size_t h_stage_n = 0;
            double *h_xstage = nullptr;      // pinned staging of the accepted models on their way out
            size_t h_xstage_n = 0;
            const double **p0tab = nullptr;  // device: where each list element's momentum row lies (chain-major)
            int64_t rows_direct = 0, rows_staged = 0;  // rows sent straight from the caller's pinned memory / gathered first
            int cap = 0;                 // list elements the device lists hold
            unsigned tag = 0, ltag = 0;
            bool dirty = false;          // an aborted launch left tags behind
            bool active[16] = {};        // chain has a trajectory in flight (carry-over mode)
            int64_t launches = 0, lock_steps = 0, lost = 0, chain_steps = 0;
            int aborts = 0;
            long long *dbg = nullptr;
        } ls;
    } rs;
    int64_t prof_res_evals = 0;

    // page-locked host memory handed to the caller (gh_pinned_alloc): momentum rows drawn into it go to the device
    // without a gather on the host
    struct Pinned {
        char *base;
        size_t bytes;
    };
    std::vector<Pinned> pinned;

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
    std::vector<int64_t> ev_bytes;  // bytes of G the timed launch of event pair i read
    int64_t prof_bytes_last = 0;
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
