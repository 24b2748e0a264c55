// pg_comm.cpp -- the ONE exchange of the locus-sharded path, inside the product: RCCL all-reduce(sum) of the partial
// kinship sums over the ranks of a node (xGMI), one rank = one pg_ctx = one GPU.
//
// Reference (gwas/ols.rs:291-295): `kinship = g.dot(&g.t()) / p` over ALL loci.  With the loci split into per-rank slabs,
// S = sum_r S_r is the only quantity that needs every rank's data (n x n doubles: 320 KB at n = 200 -- latency-bound, the
// 7 x ~153 GB/s xGMI links are never the limit, so ONE collective of the whole matrix, no bucketing); everything after it
// (eigen rule, basis, sweep of the own slab) is rank-local and deterministic, hence replicated rather than broadcast.
// The reference's parallel axis that this replaces: one worker thread per file chunk (base/sync.rs:913-939).
//
// RCCL is loaded lazily (dlopen) so that single-GPU users of libpoolgen_hip.so never map it and a process that already
// carries an RCCL (torch.distributed) shares that copy by soname.
#include "pg_common.h"
#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    std::string error;
};

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *nm : names) {
            api.handle = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
            if (api.handle) break;
        }
        if (!api.handle) { api.error = std::string("cannot load RCCL: ") + dlerror(); return; }
        auto sym = [&](const char *s) {
            void *p = dlsym(api.handle, s);
            if (!p && api.error.empty()) api.error = std::string("RCCL lacks ") + s;
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        api.GetVersion = reinterpret_cast<decltype(api.GetVersion)>(sym("ncclGetVersion"));
    });
    return api;
}

int rccl_fail(pg_ctx *ctx, ncclResult_t r, const char *what) {
    return pg_fail(ctx, PG_ERR_HIP, "%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
}

} // namespace

extern "C" int pg_comm_unique_id(void *id_out) {
    if (!id_out) return PG_ERR_INVALID;
    RcclApi &R = rccl();
    if (!R.error.empty()) return pg_fail(nullptr, PG_ERR_UNSUPPORTED, "%s", R.error.c_str());
    ncclUniqueId id;
    const ncclResult_t r = R.GetUniqueId(&id);
    if (r != ncclSuccess) return rccl_fail(nullptr, r, "ncclGetUniqueId");
    static_assert(sizeof(id) == PG_COMM_ID_BYTES, "PG_COMM_ID_BYTES must equal NCCL_UNIQUE_ID_BYTES");
    std::memcpy(id_out, &id, sizeof id);
    return PG_OK;
}

extern "C" int pg_comm_init_rank(pg_ctx *ctx, const void *id_in, int nranks, int rank) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, id_in && nranks >= 1 && rank >= 0 && rank < nranks, "comm_init_rank: bad arguments (rank %d of %d)", rank, nranks);
    if (ctx->comm) return pg_fail(ctx, PG_ERR_STATE, "comm_init_rank: this context already has a communicator");
    RcclApi &R = rccl();
    if (!R.error.empty()) return pg_fail(ctx, PG_ERR_UNSUPPORTED, "%s", R.error.c_str());
    PG_HIP(ctx, hipSetDevice(ctx->device));
    ncclUniqueId id;
    std::memcpy(&id, id_in, sizeof id);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = R.CommInitRank(&comm, nranks, id, rank);
    if (r != ncclSuccess) return rccl_fail(ctx, r, "ncclCommInitRank");
    ctx->comm = comm;
    ctx->comm_size = nranks;
    ctx->comm_rank = rank;
    return PG_OK;
}

extern "C" int pg_comm_destroy(pg_ctx *ctx) {
    if (!ctx) return PG_ERR_INVALID;
    if (!ctx->comm) return PG_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    const ncclResult_t r = rccl().CommDestroy(static_cast<ncclComm_t>(ctx->comm));
    ctx->comm = nullptr;
    ctx->comm_size = 1;
    ctx->comm_rank = 0;
    return r == ncclSuccess ? PG_OK : rccl_fail(ctx, r, "ncclCommDestroy");
}

extern "C" int pg_comm_size(const pg_ctx *ctx) { return ctx && ctx->comm ? ctx->comm_size : 1; }
extern "C" int pg_comm_rank(const pg_ctx *ctx) { return ctx && ctx->comm ? ctx->comm_rank : 0; }

extern "C" int pg_comm_version(int *version_out) {
    RcclApi &R = rccl();
    if (!R.error.empty() || !R.GetVersion) return pg_fail(nullptr, PG_ERR_UNSUPPORTED, "%s", R.error.empty() ? "RCCL lacks ncclGetVersion" : R.error.c_str());
    return R.GetVersion(version_out) == ncclSuccess ? PG_OK : PG_ERR_HIP;
}

// In-place sum over the ranks of ctx's communicator, enqueued on ctx's stream (ordered with the kernels around it).
// Without a communicator (single GPU) it is the identity.
extern "C" int pg_allreduce_sum_dev(pg_ctx *ctx, double *buf_dev, int64_t count) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, buf_dev && count > 0, "allreduce: bad arguments");
    if (!ctx->comm) return PG_OK;
    PG_HIP(ctx, hipSetDevice(ctx->device));
    pg_prof_begin(ctx, PG_K_ALLREDUCE);
    const ncclResult_t r = rccl().AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, ncclSum,
                                            static_cast<ncclComm_t>(ctx->comm), ctx->stream);
    pg_prof_end(ctx);
    if (r != ncclSuccess) return rccl_fail(ctx, r, "ncclAllReduce");
    return PG_OK;
}

// The sharded ols_iter_with_kinship of one rank: this rank's contiguous slab of loci in, this rank's
// slab of results out.  p_total = the number of columns over ALL ranks (the reference divides by it, gwas/ols.rs:295).
extern "C" int pg_ols_kinship_sharded_dev(pg_ctx *ctx, const double *G_dev, int64_t p_local, int64_t p_total, int n, int64_t ld,
                                          const double *Y, int k, double var_explained, int force_m, int *m_out, double *K_out,
                                          double *beta_dev, double *var_dev, double *pval_dev) {
    if (!ctx) return PG_ERR_INVALID;
    PG_CHECK(ctx, p_total >= p_local && p_local > 0, "ols_kinship_sharded: p_total (%lld) < p_local (%lld)", (long long)p_total,
             (long long)p_local);
    // Without a communicator the all-reduce below is the identity: a slab that is not the whole matrix would silently give
    // K = S_local / p_total and wrong fits.
    if (!ctx->comm && p_total != p_local)
        return pg_fail(ctx, PG_ERR_STATE, "ols_kinship_sharded: p_total (%lld) != p_local (%lld) on a context without a communicator "
                       "(pg_comm_init_rank first, or pass the whole matrix)", (long long)p_total, (long long)p_local);
    PG_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->S_n < n) {
        PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->S_dev) PG_HIP(ctx, hipFree(ctx->S_dev));
        ctx->S_dev = nullptr;
        ctx->S_n = 0;
        PG_HIP(ctx, hipMalloc((void **)&ctx->S_dev, sizeof(double) * n * n));
        ctx->S_n = n;
    }
    int rc = pg_set_phenotypes(ctx, force_m > 0 ? 0 : n, force_m > 0 ? nullptr : Y, force_m > 0 ? 0 : k);
    if (rc) return rc;
    rc = pg_kinship_partial_dev(ctx, G_dev, p_local, n, ld, ctx->S_dev);
    if (rc) return rc;
    rc = pg_allreduce_sum_dev(ctx, ctx->S_dev, (int64_t)n * n);
    if (rc) return rc;
    rc = pg_kinship_set(ctx, ctx->S_dev, p_total, n, Y, k, var_explained, force_m, m_out, K_out, nullptr);
    if (rc) return rc;
    return pg_ols_sweep_dev(ctx, G_dev, p_local, n, ld, beta_dev, var_dev, pval_dev);
}
