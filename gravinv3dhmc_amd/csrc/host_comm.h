// libgravhmc host side: all-reduce for a chain whose cells are sharded over GPUs (RCCL bound at run
// time, or a host callback).  Included once by gravhmc.hip.
#pragma once

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

static bool shard_cols(const gh_ctx *c) { return c->sh.kind != 0 && c->sh.axis == 0; }
static bool shard_rows(const gh_ctx *c) { return c->sh.kind != 0 && c->sh.axis == 1; }

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
    c->bt.ready = false;
    c->rs.state = 0;  // a context that ran unsharded before must not keep the resident chain kernel
    return GH_OK;
}

// Row blocks (BASELINE configs[4] as it is worded: "G row-block sharded ... RCCL reduce ... for the misfit sum";
// SURVEY 8e.2): this context holds the observations [n0, n0 + N) of N_global and ALL cells.  Model vectors are
// replicated; a column's dot with r spans the ranks, so the gradient (M doubles) is all-reduced between the
// adjoint pass and the update, and the step reads the local shard twice (adjoint pass, forward pass).
static int shard_rows_init(gh_ctx *c, int rank, int world, int64_t N_global, int64_t n0)
{
    if (world < 1 || rank < 0 || rank >= world) return fail(c, GH_ERR_ARG, "gh_shard_init_rows: bad rank/world");
    if (n0 < 0 || n0 + c->N > N_global) return fail(c, GH_ERR_ARG, "gh_shard_init_rows: observation range outside the problem");
    if (c->mf) return fail(c, GH_ERR_UNSUPPORTED, "row blocks need the stored kernel");
    if (c->wv.on) return fail(c, GH_ERR_UNSUPPORTED, "gh_shard_init_rows comes before gh_compress_wavelet (every rank compresses its own rows)");
    c->sh.axis = 1;
    c->sh.rank = rank;
    c->sh.world = world;
    c->sh.N_global = N_global;
    c->sh.n0 = n0;
    c->sh.M_global = c->M;
    c->sh.m0 = 0;
    const size_t need = (size_t)std::max<int64_t>(c->ld + 8, c->M);
    TRY(dalloc(c, &c->sh.buf, need));
    if (!c->sh.hbuf) HIPCHK(c, hipHostMalloc((void **)&c->sh.hbuf, sizeof(double) * need));
    c->sh.buf_n = need;
    TRY(dalloc(c, &c->sh.rbuf, 8));
    // (the partial sums of p'p come from vec_update_kernel: one per 256 cells)
    c->n_teams = std::max(c->n_teams, (int)((c->M + 255) / 256));
    c->chain_ready = false;
    c->bt.ready = false;
    c->rs.state = -1;  // (the resident chain kernel has no exchange between GPUs)
    return GH_OK;
}
