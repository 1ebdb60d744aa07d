// C ABI of libmsm_hip.so (declared in include/msm.h): argument checking,
// error text and precision dispatch; all device work lives in engine.hip.
#include <cstring>
#include <new>
#include <string>

#include <hip/hip_runtime.h>

#include "comm.hpp"
#include "engine.hpp"

struct msm_ctx {
    msm::IEngine* eng = nullptr;
    std::string err;
    int device = 0;
};

namespace {
thread_local std::string g_create_error;

int fail(msm_ctx* ctx, int code, const char* what) {
    if (ctx) ctx->err = what;
    return code;
}
}  // namespace

#define MSM_CHECK_CTX(ctx) \
    if (!(ctx) || !(ctx)->eng) return MSM_ERR_INVALID

namespace {
// Makes the context's GPU current for the duration of a call and puts the caller's device back afterwards: a
// process may hold contexts on several GPUs, and a thread's current device is whatever it last selected.
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

// Nothing throws across the C boundary: the engine allocates host containers inside calls, so every entry point
// body runs here.  `err` receives the text (the context's string, or a local one on threads that do not own it).
template <typename F>
int32_t guarded(const msm_ctx* ctx, std::string& err, F&& body) noexcept {
    try {
        DeviceGuard g(ctx->device);
        return body();
    } catch (const std::bad_alloc&) {
        try { err = "out of host memory"; } catch (...) {}
        return MSM_ERR_NOMEM;
    } catch (const std::exception& e) {
        try { err = e.what(); } catch (...) {}
        return MSM_ERR_INVALID;
    } catch (...) {
        try { err = "unknown exception"; } catch (...) {}
        return MSM_ERR_INVALID;
    }
}
#define MSM_CALL(ctx, expr) return guarded((ctx), (ctx)->err, [&]() -> int32_t { return (expr); })
}  // namespace

extern "C" {

int32_t msm_create(const msm_config* cfg, msm_ctx** out) {
    try {
        if (!cfg || !out) { g_create_error = "null argument"; return MSM_ERR_INVALID; }
        *out = nullptr;
        if (cfg->struct_size != (int32_t)sizeof(msm_config)) {
            g_create_error = "msm_config.struct_size does not match this library";
            return MSM_ERR_INVALID;
        }
        if (!(cfg->hbar_ > 0.0) || !(cfg->particle_mass > 0.0) || !(cfg->axis_length > 0.0)) {
            g_create_error = "axis_length, hbar_ and particle_mass must be positive";
            return MSM_ERR_INVALID;
        }
        if (cfg->precision != MSM_F64 && cfg->precision != MSM_F32) {
            g_create_error = "precision must be MSM_F64 or MSM_F32";
            return MSM_ERR_INVALID;
        }
        msm_ctx* ctx = new msm_ctx();
        ctx->device = cfg->device;
        int rc = MSM_ERR_NOMEM;
        try {
            DeviceGuard g(cfg->device);
            ctx->eng = cfg->precision == MSM_F64 ? msm::make_engine_f64(*cfg) : msm::make_engine_f32(*cfg);
            rc = ctx->eng->init(ctx->err);
        } catch (const std::bad_alloc&) {
            ctx->err = "out of host memory";
            rc = MSM_ERR_NOMEM;
        } catch (const std::exception& e) {
            ctx->err = e.what();
            rc = MSM_ERR_INVALID;
        }
        if (rc != MSM_OK) {
            g_create_error = ctx->err;
            {
                DeviceGuard g(cfg->device);
                delete ctx->eng;   // releases whatever init had allocated
            }
            delete ctx;
            return rc;
        }
        *out = ctx;
        return MSM_OK;
    } catch (const std::bad_alloc&) {
        return MSM_ERR_NOMEM;
    } catch (...) {
        return MSM_ERR_INVALID;
    }
}

int32_t msm_destroy(msm_ctx* ctx) {
    if (!ctx) return MSM_OK;
    try {
        DeviceGuard g(ctx->device);
        delete ctx->eng;
    } catch (...) {}
    delete ctx;
    return MSM_OK;
}

const char* msm_last_error(const msm_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int32_t msm_get_derived(const msm_ctx* ctx, double out[8]) {
    MSM_CHECK_CTX(ctx);
    if (!out) return MSM_ERR_INVALID;
    ctx->eng->derived(out);
    return MSM_OK;
}

int32_t msm_set_psi(msm_ctx* ctx, int32_t stream, const void* re, const void* im) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->set_psi(stream, re, im, ctx->err));
}
int32_t msm_copy_psi(msm_ctx* ctx, int32_t dst_stream, int32_t src_stream) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->copy_psi(dst_stream, src_stream, ctx->err));
}
int32_t msm_sample(msm_ctx* ctx, int32_t stream, int32_t scheme, uint64_t seed) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->sample(stream, scheme, seed, ctx->err));
}
int32_t msm_fft(msm_ctx* ctx, int32_t stream, int32_t direction) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->fft(stream, direction, ctx->err));
}
int32_t msm_potential(msm_ctx* ctx, int32_t first, int32_t count) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->potential(first, count, ctx->err));
}
int32_t msm_get_batch(const msm_ctx* ctx, int32_t* batch_streams) {
    MSM_CHECK_CTX(ctx);
    if (!batch_streams) return MSM_ERR_INVALID;
    *batch_streams = ctx->eng->batch();
    return MSM_OK;
}
int32_t msm_step(msm_ctx* ctx) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->step(ctx->err));
}
int32_t msm_step_begin(msm_ctx* ctx, const double* a_now, const double* tau_to_next_dump) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->step_begin(a_now, tau_to_next_dump, ctx->err));
}
int32_t msm_step_end(msm_ctx* ctx, const double* kick_a, const double* new_time, const double* new_tau,
                     const double* snapped_tau) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->step_end(kick_a, new_time, new_tau, snapped_tau, ctx->err));
}
int32_t msm_sync(msm_ctx* ctx) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->sync(ctx->err));
}
int32_t msm_get_status(msm_ctx* ctx, msm_stream_status* out, int32_t n) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->get_status(out, n, ctx->err));
}
int32_t msm_get_psi(msm_ctx* ctx, int32_t stream, void* re, void* im) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->get_psi(stream, re, im, false, ctx->err));
}
int32_t msm_get_psik(msm_ctx* ctx, int32_t stream, void* re, void* im) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->get_psi(stream, re, im, true, ctx->err));
}
int32_t msm_get_psi_async(msm_ctx* ctx, int32_t stream, void* re, void* im, int32_t* ticket) {
    MSM_CHECK_CTX(ctx);
    int t = -1;
    const int32_t rc = guarded(ctx, ctx->err, [&]() -> int32_t { return ctx->eng->get_psi_async(stream, re, im, &t, ctx->err); });
    if (ticket) *ticket = t;
    return rc;
}
int32_t msm_download_wait(msm_ctx* ctx, int32_t ticket) {
    MSM_CHECK_CTX(ctx);
    std::string err;   // may be called from a writer thread while the owner thread uses ctx->err
    return guarded(ctx, err, [&]() -> int32_t { return ctx->eng->download_wait(ticket, err); });
}
int32_t msm_set_clock(msm_ctx* ctx, int32_t stream, double time, double tau, int32_t current_dumps) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->set_clock(stream, time, tau, current_dumps, ctx->err));
}
int32_t msm_host_alloc(size_t bytes, void** out) {
    if (!out) return MSM_ERR_INVALID;
    return hipHostMalloc(out, bytes) == hipSuccess ? MSM_OK : MSM_ERR_NOMEM;
}
int32_t msm_host_free(void* p) { return hipHostFree(p) == hipSuccess ? MSM_OK : MSM_ERR_HIP; }
int32_t msm_get_potential(msm_ctx* ctx, int32_t stream, void* phi) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->get_potential(stream, phi, ctx->err));
}
int32_t msm_density_sum(msm_ctx* ctx, void* device_out_f64) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->density_sum(device_out_f64, ctx->err));
}
int32_t msm_step_density(msm_ctx* ctx, void* device_out_f64) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->step_density(device_out_f64, ctx->err));
}
int32_t msm_synth_accumulate(msm_ctx* ctx, int32_t stream, int32_t slot) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->synth_accumulate(stream, slot, ctx->err));
}
int32_t msm_synth_device(msm_ctx* ctx, int32_t slot, void** device_sums, int64_t* count) {
    MSM_CHECK_CTX(ctx);
    long long c = 0;
    const int32_t rc = guarded(ctx, ctx->err, [&]() -> int32_t { return ctx->eng->synth_device(slot, device_sums, &c, ctx->err); });
    if (count) *count = c;
    return rc;
}
int32_t msm_synth_set_count(msm_ctx* ctx, int32_t slot, int64_t count) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->synth_set_count(slot, count, ctx->err));
}
int32_t msm_synth_fetch(msm_ctx* ctx, int32_t slot, int32_t field, double* re, double* im) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->synth_fetch(slot, field, re, im, ctx->err));
}
int32_t msm_synth_qx(msm_ctx* ctx, int32_t slot, double* qx_re, double* qx_im) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->synth_qx(slot, qx_re, qx_im, ctx->err));
}
int32_t msm_synth_qx_noalloc(msm_ctx* ctx, int32_t slot, double* device_out2) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->synth_qx_noalloc(slot, device_out2, ctx->err));
}
int32_t msm_synth_release(msm_ctx* ctx, int32_t slot) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->synth_release(slot, ctx->err));
}
int32_t msm_comm_unique_id(void* id_out) {
    try {
        if (!id_out) return MSM_ERR_INVALID;
        std::string err;
        const msm::Rccl* r = msm::rccl(err);
        if (!r) { g_create_error = err; return MSM_ERR_INVALID; }
        msm::CommId id;
        const int rc = r->get_unique_id(&id);
        if (rc != 0) { g_create_error = std::string("ncclGetUniqueId: ") + r->error_string(rc); return MSM_ERR_HIP; }
        std::memcpy(id_out, id.bytes, msm::COMM_ID_BYTES);
        return MSM_OK;
    } catch (...) {
        return MSM_ERR_INVALID;
    }
}
const char* msm_comm_origin(void) {
    static thread_local std::string text;
    try {
        std::string err;
        const msm::Rccl* r = msm::rccl(err);
        text = r ? r->origin : err;
    } catch (...) {
        return "";
    }
    return text.c_str();
}
int32_t msm_comm_init(msm_ctx* ctx, const void* id, int32_t rank, int32_t world) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->comm_init(id, rank, world, ctx->err));
}
int32_t msm_comm_allreduce_f64(msm_ctx* ctx, double* values, int32_t n, int32_t op) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->comm_allreduce_f64(values, n, op, ctx->err));
}
int32_t msm_mean_density_begin(msm_ctx* ctx, int64_t total_streams) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->mean_density_begin(total_streams, ctx->err));
}
int32_t msm_mean_density(msm_ctx* ctx, double* host_or_null, void** device_or_null) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->mean_density(host_or_null, device_or_null, ctx->err));
}
int32_t msm_synth_allreduce(msm_ctx* ctx, int32_t slot) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->synth_allreduce(slot, ctx->err));
}

int32_t msm_timer_start(msm_ctx* ctx) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->timer_start(ctx->err));
}
int32_t msm_timer_stop_ms(msm_ctx* ctx, float* ms) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->timer_stop(ms, ctx->err));
}

int32_t msm_profile(msm_ctx* ctx, int32_t enable) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->profile(enable, ctx->err));
}
int32_t msm_profile_report(msm_ctx* ctx, int32_t max_kinds, double* total_ms, int64_t* launches, int32_t* n_kinds) {
    MSM_CHECK_CTX(ctx);
    MSM_CALL(ctx, ctx->eng->profile_report(max_kinds, total_ms, (long long*)launches, n_kinds, ctx->err));
}
const char* msm_profile_kind_name(int32_t kind) {
    static const char* names[] = {"psi_inv_z", "psi_inv_y", "psi_invx_rho", "psi_fwd_x", "psi_fwd_y", "psi_fwdz_invz",
                                  "w_fwd_y", "poisson_z", "w_inv_y", "phi_inv_x", "rho_fwd_x", "plain"};
    return (kind >= 0 && kind < 12) ? names[kind] : "";
}

}  // extern "C"
