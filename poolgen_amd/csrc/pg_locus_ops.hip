// pg_locus_ops.hip -- sync-derived per-locus operators (placeholder until the kernels land).
#include "pg_common.h"
#define PG_TODO(name) return ctx ? pg_fail(ctx, PG_ERR_UNSUPPORTED, name ": kernel not built yet") : PG_ERR_INVALID
extern "C" int pg_ols_iter_batch_dev(pg_ctx *ctx, const uint32_t *, int64_t, int, const double *, const pg_filter *, const double *, int, int32_t *, int32_t *, double *, double *, double *) { PG_TODO("pg_ols_iter_batch_dev"); }
extern "C" int pg_pearson_batch_dev(pg_ctx *ctx, const uint32_t *, int64_t, int, const double *, const pg_filter *, const double *, int, int32_t *, int32_t *, double *, double *, double *) { PG_TODO("pg_pearson_batch_dev"); }
extern "C" int pg_chisq_batch_dev(pg_ctx *ctx, const uint32_t *, int64_t, int, const double *, const pg_filter *, int32_t *, int32_t *, double *, double *) { PG_TODO("pg_chisq_batch_dev"); }
extern "C" int pg_ols_iter_batch(pg_ctx *ctx, const uint32_t *, int64_t, int, const double *, const pg_filter *, const double *, int, int32_t *, int32_t *, double *, double *, double *) { PG_TODO("pg_ols_iter_batch"); }
extern "C" int pg_pearson_batch(pg_ctx *ctx, const uint32_t *, int64_t, int, const double *, const pg_filter *, const double *, int, int32_t *, int32_t *, double *, double *, double *) { PG_TODO("pg_pearson_batch"); }
extern "C" int pg_chisq_batch(pg_ctx *ctx, const uint32_t *, int64_t, int, const double *, const pg_filter *, int32_t *, int32_t *, double *, double *) { PG_TODO("pg_chisq_batch"); }
