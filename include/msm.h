/* msm.h -- C ABI of libmsm_hip.so: the MI355X (gfx950) split-step
 * Schroedinger-Poisson evolver that stands in for the ArrayFire calls on the
 * hot path of andillio/MSM's `msm-simulator` crate.
 *
 * The reference has no FFI seam of its own on this path: every array
 * operation of `SimulationObject::update` is an ArrayFire library call.  Each
 * entry point below names the reference code it replaces (paths relative to
 * the reference checkout).  A Rust binding is a plain `extern "C"` block
 * over these signatures.
 *
 * Conventions
 *  - every function returns 0 on success, a negative msm_status otherwise;
 *    nothing throws or aborts across the boundary; msm_last_error() gives text;
 *  - one context per GPU, driven by one host thread at a time;
 *  - host buffers are caller-owned and only touched during the call;
 *  - a grid is N^dims values in C order, element [a][b][c] at a*N*N + b*N + c,
 *    which is the reference's device order (simulator/src/utils/io.rs:60-68);
 *  - "T" below is double (MSM_F64) or float (MSM_F32), chosen at create time
 *    (the reference fixes `type T = f64`, simulator/src/main.rs:19);
 *  - streams are the reference's independent simulations, one per seed
 *    (simulator/src/main.rs:43-79); this library evolves many at once.
 */
#ifndef MSM_H
#define MSM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msm_ctx msm_ctx;

enum msm_status_code {
    MSM_OK = 0,
    MSM_ERR_INVALID = -1,   /* bad argument / unsupported configuration */
    MSM_ERR_HIP = -2,       /* a HIP runtime call failed               */
    MSM_ERR_NOMEM = -3,     /* device allocation failed                */
    MSM_ERR_STATE = -4      /* call not valid in the current state     */
};

enum msm_precision { MSM_F64 = 0, MSM_F32 = 1 };
enum msm_scheme { MSM_POISSON = 0, MSM_WIGNER = 1, MSM_HUSIMI = 2 };
enum msm_direction { MSM_FORWARD = -1, MSM_INVERSE = 1 };

/* Mirrors the fields of `SimulationParameters`
 * (simulator/src/simulation_object.rs:67-140) that the device path needs. */
typedef struct msm_config {
    int32_t struct_size;        /* = sizeof(msm_config), ABI check              */
    int32_t device;             /* HIP device ordinal (reference: set_device(0)) */
    int32_t dims;               /* 1, 2 or 3 (utils/grid.rs:272-279)             */
    int32_t size;               /* N, even, in [2, 1024] (utils/fft.rs:105)      */
    int32_t precision;          /* msm_precision                                 */
    int32_t n_streams;          /* streams held by this context                  */
    int32_t batch_streams;      /* streams advanced per kernel launch; 0 = auto  */
    int32_t expanding;          /* 0 static box, 1 cosmological (cargo feature)  */
    double axis_length;
    double time;                /* start time (TOML `time`, default 0)           */
    double final_sim_time;
    double cfl;
    uint32_t num_data_dumps;
    uint32_t reserved0;
    double total_mass;
    double particle_mass;
    double hbar_;
    double k2_cutoff;
    double alias_threshold;
    /* [cosmology] (only read when expanding != 0) */
    double omega_matter_now;
    double omega_radiation_now;
    double h;
    double z0;
    double max_dloga;           /* <= 0: use the reference default 1e-3          */
    void* hip_stream;           /* hipStream_t to enqueue on; NULL = own stream  */
} msm_config;

/* Per-stream result of the last step (host copy, valid after msm_sync). */
typedef struct msm_stream_status {
    double time;            /* simulation time after the step (snapped at dumps)  */
    double dt;              /* timestep taken (d tau when expanding)              */
    double max_phi;         /* max |phi| that sized the step                      */
    double alias_mass;      /* dk^dims * sum_{k2 > cutoff k2max} |psi_k|^2        */
    double tau;             /* supercomoving time (expanding only)                */
    int64_t n_steps;
    int32_t current_dumps;
    int32_t dump;           /* 1: this step landed on a dump time                 */
    int32_t active;         /* 0 once time >= final_sim_time or aliased           */
    int32_t aliased;        /* 1: alias_mass > alias_threshold (reference panics) */
} msm_stream_status;

/* Library lifetime.  Replaces SimulationParameters::new + SimulationGrid::new
 * (simulation_object.rs:203-315): derives dx, dk = dx, hbar_, the k^2 tables
 * and k2_max, and allocates psi, psi_k and scratch for every stream. */
int32_t msm_create(const msm_config* cfg, msm_ctx** out);
int32_t msm_destroy(msm_ctx* ctx);
const char* msm_last_error(const msm_ctx* ctx);   /* ctx may be NULL: last create error */

/* Derived scalars, as the reference prints them with --verbose
 * (simulation_object.rs:325-363).  out = {dx, dk, k2_max, hbar_, n_tot,
 * kinetic_dt, comoving_boxsize(or 0), transform mode}; the last entry is
 * diagnostic: 0 = three passes per 3-D transform, 2 = a two-pass engine
 * (128^3 and 256^3 in both precisions, 512^3 in single precision; the
 * state is then held as a half-transformed image of psi internally,
 * downloads are unaffected; MSM_TWO_PASS=0 in the environment selects the
 * generic three-pass engine). */
int32_t msm_get_derived(const msm_ctx* ctx, double out[8]);

/* Initial conditions.
 * All three are only valid before the first step (afterwards the state lives in
 * Fourier space and they return MSM_ERR_STATE).
 * msm_set_psi: upload one stream's wavefunction from two host planes of T
 *   (what user_specified_ics builds from the npz members `real`/`imag`,
 *   simulator/src/ics.rs:716-726, and what every IC builder ends with).
 * msm_copy_psi: device copy stream src -> dst (every stream starts from the
 *   same mean-field psi before sampling, simulation_object.rs:406-435).
 * msm_sample: Wigner/Husimi Gaussian perturbation of one stream seeded by
 *   `seed` (sample_quantum_perturbation, ics.rs:434-648).  MSM_POISSON
 *   (ics.rs:495-558) draws the per-cell counts from the stream's own seeded
 *   Philox sequence; the reference uses a non-reproducible thread RNG there. */
int32_t msm_set_psi(msm_ctx* ctx, int32_t stream, const void* re, const void* im);
int32_t msm_copy_psi(msm_ctx* ctx, int32_t dst_stream, int32_t src_stream);
int32_t msm_sample(msm_ctx* ctx, int32_t stream, int32_t scheme, uint64_t seed);

/* Unitary FFT of one stream's psi buffer in place, scale 1/N^{dims/2} in both
 * directions (utils/fft.rs:6-98: forward/inverse/forward_inplace/inverse_inplace).
 * Only before the first step. */
int32_t msm_fft(msm_ctx* ctx, int32_t stream, int32_t direction);

/* calculate_potential (simulation_object.rs:1066-1110) for streams
 * [first, first+count): phi = Re F^-1[ -C F[M |psi|^2] / k^2 ], k = 0 mode -> 0.
 * Also refreshes max|phi| per stream (get_timestep's max_all, :905).
 * count must not exceed the batch size reported by msm_get_batch. */
int32_t msm_potential(msm_ctx* ctx, int32_t first, int32_t count);
int32_t msm_get_batch(const msm_ctx* ctx, int32_t* batch_streams);

/* One SimulationObject::update (simulation_object.rs:475-661, expanding
 * :669-873) for every active stream, asynchronous on the context's stream.
 * Static box: msm_step does the whole update.
 * Expanding: msm_step_begin runs the potential + timestep part (needs the
 *   scale factor a and tau-to-next-dump per stream), the caller advances its
 *   scale-factor solver with the returned d tau, then msm_step_end applies the
 *   rest with kick_a[s] = a1 + a2 (the two half-kicks share phi, :726-742) and
 *   the new (time, tau) per stream. */
int32_t msm_step(msm_ctx* ctx);
int32_t msm_step_begin(msm_ctx* ctx, const double* a_now, const double* tau_to_next_dump);
int32_t msm_step_end(msm_ctx* ctx, const double* kick_a, const double* new_time, const double* new_tau,
                     const double* snapped_tau);

/* Wait for enqueued work; then status[0..n_streams) is current. */
int32_t msm_sync(msm_ctx* ctx);
int32_t msm_get_status(msm_ctx* ctx, msm_stream_status* out, int32_t n);

/* Downloads (replace Array::host in complex_array_to_disk, utils/io.rs:46-47).
 * psi / psi_k: two planes of T; after the first step psi is rebuilt as
 * F^-1[psi_k] (simulation_object.rs:581) for the download.  potential: one
 * plane of T, the kick potential of the most recent msm_potential / step batch. */
int32_t msm_get_psi(msm_ctx* ctx, int32_t stream, void* re, void* im);
int32_t msm_get_psik(msm_ctx* ctx, int32_t stream, void* re, void* im);
int32_t msm_get_potential(msm_ctx* ctx, int32_t stream, void* phi);

/* Dump pipeline.  msm_get_psi_async forms the planes on the compute stream and copies them out
 * on a second stream (one of four staging grids per context); the call returns at once and the
 * steps that follow overlap the transfer.  `re`/`im` should be pinned (msm_host_alloc) for the copy
 * to be asynchronous; they are complete after msm_download_wait(ticket), which may be called from
 * another thread (the dump writers of utils/io.rs:58-85).
 * msm_set_clock: restart -- before the first step, place a stream at (time, tau) with
 * `current_dumps` dumps already written, e.g. after msm_set_psi from its last dump.  The reference
 * has no counterpart: its dumps are write-only (simulation_object.rs:627,915 are TODOs). */
int32_t msm_get_psi_async(msm_ctx* ctx, int32_t stream, void* re, void* im, int32_t* ticket);
int32_t msm_download_wait(msm_ctx* ctx, int32_t ticket);
int32_t msm_host_alloc(size_t bytes, void** out);
int32_t msm_host_free(void* p);
int32_t msm_set_clock(msm_ctx* ctx, int32_t stream, double time, double tau, int32_t current_dumps);

/* sum over this context's streams of |psi_s|^2 into a DEVICE buffer of N^dims
 * doubles (caller-owned; e.g. a torch tensor that is then all-reduced over
 * RCCL).  In-situ form of the synthesizer's psi2 accumulation
 * (synthesizer/src/lib.rs:217-229); it never feeds back into the evolution. */
int32_t msm_density_sum(msm_ctx* ctx, void* device_out_f64);
/* Fast path of the same product: register a DEVICE buffer of N^dims doubles (NULL to stop);
 * every following step zeroes it and leaves in it the sum of |psi_s(t + dt)|^2 over the
 * streams that took the step (one that aliased in it is left out), accumulated inside the step's last pass at no extra HBM read (3-D
 * power-of-two grids; the other grids add one read of the batch).  Precision: the sum is formed in double from each
 * stream's components widened to double -- except in the single-precision 512^3 tiling, where a stream's |psi|^2
 * crosses memory as a single-precision grid first, so every term carries a relative rounding error of about 6e-8
 * before it is added in double (the same holds for msm_mean_density there). */
int32_t msm_step_density(msm_ctx* ctx, void* device_out_f64);

/* In-situ synthesizer (synthesizer/src/main.rs:63-93,161-187; lib.rs:190-300): the per-dump
 * ensemble products the reference's post-processor builds by re-reading every stream's dump from
 * disk.  A slot holds four running sums over streams, each N^dims complex doubles, in the dump
 * file's element order:  psi,  |psi|^2,  psik,  |psik|^2,  where psik is the UN-NORMALISED forward
 * transform of psi, as the synthesizer's ndrustfft computes it (= N^{dims/2} times the simulator's
 * unitary psi_k).
 * msm_synth_accumulate: add stream `stream`'s current psi to slot `slot` (slots are created zeroed
 *   on first use; one slot per dump index in flight).
 * msm_synth_device: the slot's device buffer (4 * N^dims complex doubles, sums not means) and the
 *   number of streams accumulated -- for a caller that all-reduces both over its GPUs; after such a
 *   reduction msm_synth_set_count tells the library the global count.
 * msm_synth_fetch: mean field `field` (MSM_SYNTH_PSI ...) to two host planes of doubles.
 * msm_synth_qx: Qx = sum( <|psi|^2> - |<psi>|^2 ) * dv  (main.rs:161-175), dv = dx^dims.
 * msm_synth_release: free the slot. */
enum msm_synth_field { MSM_SYNTH_PSI = 0, MSM_SYNTH_PSI2 = 1, MSM_SYNTH_PSIK = 2, MSM_SYNTH_PSIK2 = 3 };
int32_t msm_synth_accumulate(msm_ctx* ctx, int32_t stream, int32_t slot);
int32_t msm_synth_device(msm_ctx* ctx, int32_t slot, void** device_sums, int64_t* count);
int32_t msm_synth_set_count(msm_ctx* ctx, int32_t slot, int64_t count);
int32_t msm_synth_fetch(msm_ctx* ctx, int32_t slot, int32_t field, double* re, double* im);
int32_t msm_synth_qx(msm_ctx* ctx, int32_t slot, double* qx_re, double* qx_im);
int32_t msm_synth_release(msm_ctx* ctx, int32_t slot);
/* msm_synth_qx for callers that cannot allocate (code under HIP graph capture, tight monitoring loops,
 * harnesses that forbid hipMalloc); it has no counterpart in the reference.  Same value as msm_synth_qx,
 * bit for bit, but nothing is allocated, freed, copied to the host or waited for: the reduction's partial
 * sums live in a fixed 16-KiB workspace inside the library (one copy per GPU and precision -- the MSM_F64 and
 * MSM_F32 engines each hold their own -- kept until the process ends),
 * and {Re Qx, Im Qx} is written to `device_out2`, two doubles the GPU can address (device memory, or pinned
 * host memory from msm_host_alloc), when the context's stream reaches the call (msm_sync to wait).
 * Limit: grids of at most 2^30 cells (1024^3); larger ones are rejected with MSM_ERR_INVALID.
 * The workspace is overwritten completely by every call, so nothing has to be cleared, but it is shared by
 * every context of the same precision on the same GPU: such calls that overlap in time on one GPU (different
 * streams) race. */
int32_t msm_synth_qx_noalloc(msm_ctx* ctx, int32_t slot, double* device_out2);

/* Several GPUs: one process and one context per GPU, the seed streams partitioned by the caller
 * (they are independent simulations, simulator/src/main.rs:43-79, so no exchange is needed to evolve
 * them).  What crosses GPUs are the ensemble products the reference's post-processor builds from the
 * dumps (synthesizer/src/lib.rs:217-229,285-306; its own MPI gathers: lib.rs:466-583): here they are
 * summed in situ by RCCL all-reduces over xGMI.  librccl is bound at run time on the first call.
 * msm_comm_unique_id: 128 bytes (an ncclUniqueId) that one rank creates and hands to all others by any
 *   means (file, environment, a launcher's key-value store).
 * msm_comm_init: joins the communicator; collective, every rank passes the same id and world.
 * msm_comm_allreduce_f64: n <= 64 host doubles reduced over the ranks in place (op 0 sum, 2 max, 3 min;
 *   RCCL's codes), synchronous -- the flags and clocks a multi-rank driver agrees on.
 * msm_mean_density_begin: from now on every step (msm_step / msm_step_end) also leaves
 *   sum over ALL ranks' streams of |psi_s(t + dt)|^2: the local sum is formed inside the step's last
 *   pass, its all-reduce is enqueued on a second stream behind an event and overlaps the next step
 *   (two buffers alternate).  Collective: every rank must then take the same number of steps.
 *   The product never feeds back into any stream's potential.  total_streams = the divisor of the mean.
 * msm_mean_density: waits for the all-reduce of the most recent step and returns the mean density
 *   <|psi|^2> as N^dims doubles: copied to `host_or_null` if given, and/or as a device pointer that stays
 *   valid until the next call.
 * msm_synth_allreduce: sums slot `slot` of the in-situ synthesizer (fields and stream count) over the
 *   ranks; collective, a rank that has accumulated nothing contributes zeros.  Afterwards
 *   msm_synth_fetch / msm_synth_qx on any rank return the ensemble's products.
 * ORDER RULE.  A context has one communicator and enqueues every collective on one communication stream of
 * its own, in the order the host issues the calls (the compute stream meets it through events).  RCCL matches
 * the n-th collective of a communicator on one rank with the n-th on every other rank, so all ranks must issue
 * the collective calls of their contexts -- msm_comm_init, msm_comm_allreduce_f64, msm_synth_allreduce, and,
 * once msm_mean_density_begin is on, every msm_step / msm_step_end -- in the same host order and the same
 * number of times; a rank that issues one more, one fewer, or two in the other order leaves all ranks waiting
 * in the collective.  msm_mean_density and the other calls are local.
 * msm_comm_origin: which shared object the RCCL entry points were bound from ("librccl.so.1 (already loaded)",
 *   a path, ...), or the reason none could be bound; diagnostic text for a launcher's per-rank log.  Binds librccl
 *   if no earlier call has.
 */
#define MSM_COMM_ID_BYTES 128
int32_t msm_comm_unique_id(void* id_out);
const char* msm_comm_origin(void);
int32_t msm_comm_init(msm_ctx* ctx, const void* id, int32_t rank, int32_t world);
int32_t msm_comm_allreduce_f64(msm_ctx* ctx, double* values, int32_t n, int32_t op);
int32_t msm_mean_density_begin(msm_ctx* ctx, int64_t total_streams);
int32_t msm_mean_density(msm_ctx* ctx, double* host_or_null, void** device_or_null);
int32_t msm_synth_allreduce(msm_ctx* ctx, int32_t slot);

/* Timing helper for benchmarks: HIP events on the context's stream. */
int32_t msm_timer_start(msm_ctx* ctx);
int32_t msm_timer_stop_ms(msm_ctx* ctx, float* ms);

/* Per-launch timing of the pass kernels with HIP events on the launch stream
 * (no counterpart in the reference; measurement hook for benchmarks).
 * msm_profile(ctx, 1) starts recording, msm_profile_report sums elapsed ms and
 * launch counts per pass kind (msm_profile_kind_name) and clears the records.
 * max_kinds must be >= 16. */
int32_t msm_profile(msm_ctx* ctx, int32_t enable);
int32_t msm_profile_report(msm_ctx* ctx, int32_t max_kinds, double* total_ms, int64_t* launches, int32_t* n_kinds);
const char* msm_profile_kind_name(int32_t kind);

#ifdef __cplusplus
}
#endif
#endif /* MSM_H */
