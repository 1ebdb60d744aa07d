// Type-erased engine interface between the C ABI (api.cpp) and the
// per-precision HIP engines (engine.hip compiled once per real type).
#pragma once
#include <cstdint>
#include <string>
#include "../../include/msm.h"

namespace msm {

struct IEngine {
    virtual ~IEngine() {}
    virtual int init(std::string& err) = 0;
    virtual int set_psi(int s, const void* re, const void* im, std::string& err) = 0;
    virtual int copy_psi(int dst, int src, std::string& err) = 0;
    virtual int sample(int s, int scheme, uint64_t seed, std::string& err) = 0;
    virtual int fft(int s, int dir, std::string& err) = 0;
    virtual int potential(int first, int count, std::string& err) = 0;
    virtual int step(std::string& err) = 0;
    virtual int step_begin(const double* a_now, const double* tau_next, std::string& err) = 0;
    virtual int step_end(const double* kick_a, const double* new_time, const double* new_tau,
                         const double* snapped_tau, std::string& err) = 0;
    virtual int sync(std::string& err) = 0;
    virtual int get_status(msm_stream_status* out, int n, std::string& err) = 0;
    virtual int get_psi(int s, void* re, void* im, bool kspace, std::string& err) = 0;
    virtual int get_psi_async(int s, void* re, void* im, int* ticket, std::string& err) = 0;
    virtual int download_wait(int ticket, std::string& err) = 0;
    virtual int set_clock(int s, double time, double tau, int current_dumps, std::string& err) = 0;
    virtual int get_potential(int s, void* phi, std::string& err) = 0;
    virtual int density_sum(void* dev_out, std::string& err) = 0;
    virtual int step_density(void* dev_out, std::string& err) = 0;
    virtual int synth_accumulate(int s, int slot, std::string& err) = 0;
    virtual int synth_device(int slot, void** dev, long long* count, std::string& err) = 0;
    virtual int synth_set_count(int slot, long long count, std::string& err) = 0;
    virtual int synth_fetch(int slot, int field, double* re, double* im, std::string& err) = 0;
    virtual int synth_qx(int slot, double* re, double* im, std::string& err) = 0;
    virtual int synth_qx_noalloc(int slot, double* out2, std::string& err) = 0;
    virtual int synth_release(int slot, std::string& err) = 0;
    virtual int comm_init(const void* id, int rank, int world, std::string& err) = 0;
    virtual int mean_density_begin(long long total_streams, std::string& err) = 0;
    virtual int mean_density(double* host, void** dev, std::string& err) = 0;
    virtual int synth_allreduce(int slot, std::string& err) = 0;
    virtual int comm_allreduce_f64(double* vals, int n, int op, std::string& err) = 0;
    virtual int timer_start(std::string& err) = 0;
    virtual int timer_stop(float* ms, std::string& err) = 0;
    virtual int profile(int enable, std::string& err) = 0;
    virtual int profile_report(int max_kinds, double* total_ms, long long* launches, int* n_kinds,
                               std::string& err) = 0;
    virtual void derived(double out[8]) const = 0;
    virtual int batch() const = 0;
};

IEngine* make_engine_f64(const msm_config& cfg);
IEngine* make_engine_f32(const msm_config& cfg);

}  // namespace msm
