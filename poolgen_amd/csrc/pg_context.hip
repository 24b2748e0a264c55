// pg_context.hip -- context lifetime, error reporting, workspace and HIP-event profiling.
#include "pg_common.h"

int pg_fail(pg_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

static std::string g_create_error;

extern "C" const char *pg_version(void) { return "poolgen_hip 0.1.0 (gfx950)"; }

extern "C" const char *pg_last_error(const pg_ctx *ctx) {
    return ctx ? ctx->err.c_str() : g_create_error.c_str();
}

extern "C" int pg_create(pg_ctx **out, int device, void *stream) {
    if (!out) return PG_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0) {
        g_create_error = std::string("no HIP device available: ") + hipGetErrorString(e) +
                         " -- libpoolgen_hip has no CPU fallback";
        return PG_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= count) {
        g_create_error = "device ordinal out of range";
        return PG_ERR_INVALID;
    }
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipGetDeviceProperties: ") + hipGetErrorString(e);
        return PG_ERR_HIP;
    }
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("device is ") + prop.gcnArchName +
                         ", this library contains gfx950 code objects only";
        return PG_ERR_NO_DEVICE;
    }
    e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return PG_ERR_HIP;
    }
    pg_ctx *ctx = new pg_ctx();
    ctx->device = device;
    ctx->cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    // NULL = the device's default (null) stream, which is what torch hands out as cuda_stream 0;
    // work is therefore always ordered with the caller's stream.
    ctx->stream = (hipStream_t)stream;
    ctx->own_stream = false;
    *out = ctx;
    return PG_OK;
}

extern "C" void pg_destroy(pg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->comm) (void)pg_comm_destroy(ctx);
    for (auto &p : ctx->ev_pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto &p : ctx->ev_free) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    if (ctx->ws) (void)hipFree(ctx->ws);
    if (ctx->W_dev) (void)hipFree(ctx->W_dev);
    if (ctx->S_dev) (void)hipFree(ctx->S_dev);
    if (ctx->lz_dev) (void)hipFree(ctx->lz_dev);
    if (ctx->ph_ytil_dev) (void)hipFree(ctx->ph_ytil_dev);
    if (ctx->spec_dev) (void)hipFree(ctx->spec_dev);
    if (ctx->syy_dev) (void)hipFree(ctx->syy_dev);
    if (ctx->tcoef_dev) (void)hipFree(ctx->tcoef_dev);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int pg_synchronize(pg_ctx *ctx) {
    if (!ctx) return PG_ERR_INVALID;
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PG_OK;
}

int pg_ws_reserve(pg_ctx *ctx, size_t bytes) {
    ctx->load_valid = false; // whoever asks for the workspace is about to overwrite it
    if (bytes <= ctx->ws_bytes) return PG_OK;
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) PG_HIP(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr;
    ctx->ws_bytes = 0;
    PG_HIP(ctx, hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return PG_OK;
}

int pg_pin_reserve(pg_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->pin_bytes) return PG_OK;
    PG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->pin) PG_HIP(ctx, hipHostFree(ctx->pin));
    ctx->pin = nullptr;
    ctx->pin_bytes = 0;
    PG_HIP(ctx, hipHostMalloc(&ctx->pin, bytes, hipHostMallocDefault));
    ctx->pin_bytes = bytes;
    return PG_OK;
}

// ---- profiling: one event pair per launch, on the launch stream, resolved at query time ----
void pg_prof_begin(pg_ctx *ctx, int kid) {
    if (!ctx->prof || kid < 0) return;
    pg_event_pair p;
    if (!ctx->ev_free.empty()) {
        p = ctx->ev_free.back();
        ctx->ev_free.pop_back();
    } else {
        if (hipEventCreate(&p.a) != hipSuccess) return;
        if (hipEventCreate(&p.b) != hipSuccess) { (void)hipEventDestroy(p.a); return; }
    }
    p.kid = kid;
    (void)hipEventRecord(p.a, ctx->stream);
    ctx->ev_pending.push_back(p);
}

void pg_prof_end(pg_ctx *ctx) {
    if (!ctx->prof || ctx->ev_pending.empty()) return;
    (void)hipEventRecord(ctx->ev_pending.back().b, ctx->stream);
}

static void prof_drain(pg_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->ev_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            ctx->prof_ms[p.kid & 0xff] += ms;
            if (!(p.kid & PG_PROF_CONT)) ctx->prof_n[p.kid & 0xff] += 1; // (a continuation bracket adds time to the launch its first bracket counted)
        }
        ctx->ev_free.push_back(p);
    }
    ctx->ev_pending.clear();
}

extern "C" int pg_profile_enable(pg_ctx *ctx, int on) {
    if (!ctx) return PG_ERR_INVALID;
    prof_drain(ctx);
    ctx->prof = on != 0;
    return PG_OK;
}

extern "C" int pg_profile_reset(pg_ctx *ctx) {
    if (!ctx) return PG_ERR_INVALID;
    prof_drain(ctx);
    for (int i = 0; i < PG_K_COUNT; ++i) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; }
    return PG_OK;
}

extern "C" int pg_profile_get(pg_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches) {
    if (!ctx || kernel_id < 0 || kernel_id >= PG_K_COUNT) return PG_ERR_INVALID;
    prof_drain(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[kernel_id];
    if (launches) *launches = ctx->prof_n[kernel_id];
    return PG_OK;
}
