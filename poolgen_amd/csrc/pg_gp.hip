// pg_gp.hip -- gp::ols (placeholder until the kernels land).
#include "pg_common.h"
extern "C" int pg_gp_ols_dev(pg_ctx *ctx, const double *, int64_t, int, int64_t, const double *, int, const int64_t *, int, const double *, double *) {
    return ctx ? pg_fail(ctx, PG_ERR_UNSUPPORTED, "pg_gp_ols_dev: kernel not built yet") : PG_ERR_INVALID;
}
