// Split-step Schroedinger-Poisson engine for gfx950, one real type per
// translation unit (-DMSM_REAL=double|float -DMSM_SUFFIX=f64|f32).
//
// Device layout (per context):
//   P[n_streams][N^3]  complex  the state: psi before the first step; afterwards psi_k
//                               (simulation_object.rs:57,60) -- in the 256^3 engine its
//                               half-inverted image S = pass A of psi, see kernels_two_pass.inc
//   X[batch][N^3]      complex  psi of the current batch inside a step (scratch)
//   W[batch/2][N^3]    complex  Poisson work array: two streams' densities
//                               packed as re/im of one complex grid
//   PHI[batch][N^3]    real     potential per stream     simulation_object.rs:63
// Three implementations of the same step share this file's host side:
//   kernels_generic.inc   any power-of-two N <= 512: a 3-D transform is three passes of the batched line FFT
//                         (fft_core.hpp), every pointwise operator applied inside a pass's load or store,
//                         neighbouring passes on the same lines fused
//   kernels_two_pass.inc  N = 256, 3-D: two passes per transform through a blocked intermediate layout
//   kernels_unfused.inc   1-D / 2-D grids and edges that are not a power of two: whole transforms with one
//                         kernel per pointwise operator (chirp-z lines for the odd sizes)
// kernels_scalar.inc holds the per-stream bookkeeping kernels, sampling and reductions.
// Between steps only one complex grid per stream is kept; psi is rebuilt on demand for downloads.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>

#include "comm.hpp"
#include "engine.hpp"
#include "fft_core.hpp"

#ifndef MSM_REAL
#define MSM_REAL double
#endif

namespace msm {
namespace {

using real = MSM_REAL;
using cplx = cx<real>;

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t e__ = (expr);                                                   \
        if (e__ != hipSuccess) {                                                   \
            err = std::string(#expr) + ": " + hipGetErrorString(e__);              \
            return e__ == hipErrorOutOfMemory ? MSM_ERR_NOMEM : MSM_ERR_HIP;       \
        }                                                                          \
    } while (0)

// ---------------------------------------------------------------------------
// device-side per-stream state (structure of arrays) and step constants
// ---------------------------------------------------------------------------
struct State {
    real* time;
    real* tau;
    real* dt;
    real* kick;                     // phase per unit phi of the kick: -dt/hbar_ (static)
    real* a_now;                    // expanding: scale factor at step start
    real* tau_next;                 // expanding: tau to next dump
    unsigned long long* maxphi;     // bits of (double) max|phi| of psi(t), >= 0
    double* maxphi_step;            // the value that sized the last timestep
    double* alias_sum;
    int* active;
    int* dump;
    int* dumps;
    int* aliased;
    int* stepped;                   // took the step just finished and did not alias: mask of the step's tail
    long long* nsteps;
    cplx* drift;                    // [n_streams][N] exp(i c k_i^2 (2 pi)^2)
    msm_stream_status* status;      // [n_streams]
};

struct Consts {
    real cfl, box, hbar, k2max, tfinal, dkpow, alias_thr, alias_cut, mass_factor, pois_c, fourpi2;
    real inv_n3;      // 1 / N^dims : forward and inverse unitary scales combined
    real unit_scale;  // 1 / N^(dims/2)
    int ndumps, expanding, N, dims;
};

// ---------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------
// sin/cos of a kick phase in double: three-term Cody-Waite reduction by pi/2 (33-bit pieces, the
// products k*piece are exact for |k| < 2^20) and the fdlibm minimax kernels on [-pi/4, pi/4].
// About 1 ulp for |x| < 1e6; beyond that the error grows like the spacing of doubles at x.
// Far fewer registers and instructions than the general-purpose library routine, which carries a
// Payne-Hanek path the bounded kick phase never needs.
__device__ __forceinline__ void sincos_t(double x, double* s, double* c) {
    const double k = rint(x * 6.36619772367581382433e-01);
    double r = fma(-k, 1.57079632673412561417e+00, x);
    r = fma(-k, 6.07710050630396597660e-11, r);
    r = fma(-k, 2.02226624871116645580e-21, r);
    r = fma(-k, 8.47842766036889956997e-32, r);
    const double z = r * r;
    double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = fma(z, ps, 2.75573137070700676789e-06);
    ps = fma(z, ps, -1.98412698298579493134e-04);
    ps = fma(z, ps, 8.33333333332248946124e-03);
    ps = fma(z, ps, -1.66666666666666324348e-01);
    const double sn = fma(z * r, ps, r);
    double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = fma(z, pc, -2.75573143513906633035e-07);
    pc = fma(z, pc, 2.48015872894767294178e-05);
    pc = fma(z, pc, -1.38888888888741095749e-03);
    pc = fma(z, pc, 4.16666666666666019037e-02);
    const double hz = 0.5 * z;
    const double w1 = 1.0 - hz;
    const double cs = w1 + (((1.0 - w1) - hz) + z * z * pc);
    const int q = (int)k & 3;
    const double ss = (q & 1) ? cs : sn, cc = (q & 1) ? sn : cs;
    *s = (q & 2) ? -ss : ss;
    *c = ((q + 1) & 2) ? -cc : cc;
}
__device__ __forceinline__ void sincos_t(float x, float* s, float* c) { sincosf(x, s, c); }

// max with NaN winning, so that a poisoned potential stays visible
__device__ __forceinline__ double nanmax(double a, double b) {
    const double m = __builtin_fmax(a, b);
    return __builtin_isunordered(a, b) ? __builtin_nan("") : m;   // one unordered compare and a select, no branches
}
// Branch-free accumulation of max|v| over many values: the running maximum ignores NaNs, a flag remembers them,
// absmax_done puts the NaN back.  (A chain of nanmax compiles to nested exec-mask branches, two per value.)
__device__ __forceinline__ void absmax_acc(double& m, bool& bad, double v) {
    m = __builtin_fmax(m, __builtin_fabs(v));
    bad |= (v != v);
}
__device__ __forceinline__ double absmax_done(double m, bool bad) { return bad ? __builtin_nan("") : m; }
// (single precision: the maximum of magnitudes is exact in the values' own type; it is widened once, by absmax_done)
__device__ __forceinline__ void absmax_acc(float& m, bool& bad, float v) {
    m = __builtin_fmaxf(m, __builtin_fabsf(v));
    bad |= (v != v);
}

// Workgroup reductions in a fixed order (bitwise reproducible from run to run): whole wavefronts reduce by
// lane exchange and meet in one LDS round, smaller workgroups walk an LDS tree.  The barriers order LDS traffic only,
// so loads and stores the caller has in flight stay in flight.  `red` holds at least THREADS doubles.
template <int THREADS, typename OP>
__device__ __forceinline__ double block_reduce(double v, double* red, OP op) {
    const int tid = threadIdx.x;
    if constexpr (THREADS % 64 == 0) {
        constexpr int WAVES = THREADS / 64;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = op(v, __shfl_xor(v, off));
        if constexpr (WAVES == 1) return v;
        if ((tid & 63) == 0) red[tid >> 6] = v;
        lds_barrier();
        double r = red[0];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) r = op(r, red[w]);
        return r;
    } else {
        red[tid] = v;
        lds_barrier();
#pragma unroll
        for (int off = THREADS / 2; off > 0; off >>= 1) {
            if (tid < off) red[tid] = op(red[tid], red[tid + off]);
            lds_barrier();
        }
        return red[0];
    }
}
template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double* red) {
    return block_reduce<THREADS>(v, red, [](double a, double b) { return a + b; });
}
// NaN must win so that a poisoned potential is visible to the host
template <int THREADS>
__device__ __forceinline__ double block_max(double v, double* red) {
    return block_reduce<THREADS>(v, red, [](double a, double b) { return nanmax(a, b); });
}

// 1/x for the Green's function: hardware reciprocal plus one Newton step (about 1 ulp in double),
// a third of the instructions of the IEEE division expansion.
__device__ __forceinline__ double recip_t(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ float recip_t(float x) {      // the same in single precision (the IEEE division expands to a dozen instructions)
    const float r = __builtin_amdgcn_rcpf(x);
    return fmaf(fmaf(-x, r, 1.0f), r, r);
}

// Hides a lane offset from common-subexpression elimination: addresses formed from the result are
// rebuilt where they are used instead of being kept (or spilled) since their first use.
__device__ __forceinline__ unsigned fresh(unsigned v) { asm volatile("" : "+v"(v)); return v; }

// Buffer addressing for the tile loads and stores whose sixteen accesses lie far apart: descriptor
// (wave-uniform base) + one 32-bit lane offset + a scalar offset per access, all in bytes and below
// 4 GiB in total.  No 64-bit address pair per access has to live in vector registers.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef double f64x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
using bufdesc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ bufdesc buf_of(const void* p) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ void buf_ld(double2& v, bufdesc r, unsigned lane_bytes, unsigned uni_bytes) {
    const f64x2 d = __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(r, (int)lane_bytes, (int)uni_bytes, 0));
    v.x = d.x; v.y = d.y;
}
__device__ __forceinline__ void buf_ld(float2& v, bufdesc r, unsigned lane_bytes, unsigned uni_bytes) {
    const f32x2 d = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(r, (int)lane_bytes, (int)uni_bytes, 0));
    v.x = d.x; v.y = d.y;
}
__device__ __forceinline__ void buf_st(double2 v, bufdesc r, unsigned lane_bytes, unsigned uni_bytes) {
    f64x2 d; d.x = v.x; d.y = v.y;
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, d), r, (int)lane_bytes, (int)uni_bytes, 0);
}
__device__ __forceinline__ void buf_st(float2 v, bufdesc r, unsigned lane_bytes, unsigned uni_bytes) {
    f32x2 d; d.x = v.x; d.y = v.y;
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, d), r, (int)lane_bytes, (int)uni_bytes, 0);
}

// max of two values per thread over the workgroup in one pass: wavefront shuffles, then one LDS
// round over the (at most 16) wavefronts.  NaN wins so that a poisoned potential stays visible.
template <int THREADS>
__device__ __forceinline__ void block_max2(double& a, double& b, double* red) {
    constexpr int WAVES = (THREADS + 63) / 64;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        a = nanmax(a, __shfl_xor(a, off));
        b = nanmax(b, __shfl_xor(b, off));
    }
    if constexpr (WAVES > 1) {
        const int wave = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[2 * wave] = a; red[2 * wave + 1] = b; }
        lds_barrier();   // LDS only: a look-ahead fetch of the caller stays in flight across the reduction
        a = red[0]; b = red[1];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) { a = nanmax(a, red[2 * w]); b = nanmax(b, red[2 * w + 1]); }
    }
}

// Tile geometry shared by all pass kernels.  AX = 0: lines along x (unit
// stride); 1: along y (stride N); 2: along z (stride N^2).  c1/c2 are the two
// coordinates that are constant along the line, in (x,y,z) order with the pass
// axis removed.
template <int N, int AX>
struct Tile {
    using G = Geom<N, AX>;
    int ln, j0, c1, c2;
    size_t base, stride;
    __device__ __forceinline__ Tile(int tile, int tid) {
        constexpr int LPW = G::LPW, TPL = G::TPL;
        if constexpr (AX == 0) {
            j0 = tid % TPL;
            ln = tid / TPL;
            const int line = tile * LPW + ln;
            base = (size_t)line * N;
            stride = 1;
            c1 = line % N;  // y
            c2 = line / N;  // z
        } else if constexpr (AX == 1) {
            ln = tid % LPW;
            j0 = tid / LPW;
            constexpr int TPP = N / LPW;
            const int z = tile / TPP;
            const int x = (tile % TPP) * LPW + ln;
            base = (size_t)z * N * N + x;
            stride = N;
            c1 = x;
            c2 = z;
        } else {
            ln = tid % LPW;
            j0 = tid / LPW;
            const int m = tile * LPW + ln;
            base = (size_t)m;
            stride = (size_t)N * N;
            c1 = m % N;  // x
            c2 = m / N;  // y
        }
    }
};

// The same in two halves for kernels whose first tile loads can go out in between: the table entry's load is then the
// oldest one in flight and its LDS write waits for it alone (loads retire in order), instead of the tile loads waiting
// for the table.  One entry per thread: every pass workgroup has at least N threads.
template <int N, int AX>
__device__ __forceinline__ cplx tw_fetch(const cplx* __restrict__ tw_g) {
    static_assert(!Geom<N, AX>::USES_LDS || Geom<N, AX>::THREADS >= N, "one table entry per thread");
    if constexpr (Geom<N, AX>::USES_LDS) return tw_g[TwLayout<N>::source(threadIdx.x & (N - 1))];
    else return mk<real>(0, 0);
}
template <int N, int AX>
__device__ __forceinline__ void tw_publish(cplx* tw_lds, cplx v) {
    if constexpr (Geom<N, AX>::USES_LDS) {
        if ((int)threadIdx.x < N) tw_lds[threadIdx.x] = v;
        if constexpr (AX == 0) lds_barrier();   // as in load_twiddles
    }
}
template <int N, int AX>
__device__ __forceinline__ void load_twiddles(cplx* tw_lds, const cplx* __restrict__ tw_g) {
    using G = Geom<N, AX>;
    if constexpr (G::USES_LDS) {
        for (int i = threadIdx.x; i < N; i += G::THREADS) tw_lds[i] = tw_g[TwLayout<N>::source(i)];
        // visibility: x lines exchange inside their wavefront, so the table needs its own barrier;
        // on the strided axes the first use is after the stage-0 exchange barrier
        if constexpr (AX == 0) lds_barrier();
    }
}

#include "kernels_generic.inc"
#include "kernels_two_pass.inc"
#include "kernels_two_pass_512.inc"
#include "kernels_two_pass_128.inc"
#include "kernels_scalar.inc"
#include "kernels_unfused.inc"

// ---------------------------------------------------------------------------
// Fixed workspace of the entry point that allocates nothing (msm_synth_qx_noalloc): the per-workgroup partial sums
// of the Qx reduction.  Every GPU this library runs on holds its own copy of the array until the process ends
// (16 KiB per precision).  Host code never uses the array's name as an address: qx_scratch_of() asks the runtime
// for the copy's device address, once per device with that device current, and everything else uses that pointer.
// ---------------------------------------------------------------------------
constexpr int QX_PARTS = 1024;                       // workgroups of k_synth_qx = partial sums
constexpr size_t QX_MAX_CELLS = (size_t)1 << 30;     // 1024^3, the largest grid the library builds
static __device__ __align__(256) double2 scratch[QX_PARTS];

constexpr int QX_MAX_DEVICES = 64;
double2* g_qx_scratch[QX_MAX_DEVICES] = {};
std::mutex g_qx_mutex;
// device address of `scratch` on the CURRENT device `dev`; null + err on failure
double2* qx_scratch_of(int dev, std::string& err) {
    if (dev < 0 || dev >= QX_MAX_DEVICES) { err = "device ordinal beyond the scratch table"; return nullptr; }
    std::lock_guard<std::mutex> lk(g_qx_mutex);
    if (!g_qx_scratch[dev]) {
        double2* ptr = nullptr;
        const hipError_t e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !ptr) { err = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e); return nullptr; }
        g_qx_scratch[dev] = ptr;
    }
    return g_qx_scratch[dev];
}
// sums the partials in index order (the order msm_synth_qx's host loop uses) and scales by dv
__global__ void k_qx_finish(const double2* __restrict__ part, int n, double dv, double* __restrict__ out2) {
#pragma clang fp contract(off)
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    double sr = 0.0, si = 0.0;
    for (int i = 0; i < n; ++i) { sr += part[i].x; si += part[i].y; }
    out2[0] = sr * dv;
    out2[1] = si * dv;
}

// ---------------------------------------------------------------------------
// size dispatch
// ---------------------------------------------------------------------------
template <typename F>
bool dispatch_n(int N, F&& f) {
    switch (N) {
        case 2: f(std::integral_constant<int, 2>{}); return true;
        case 4: f(std::integral_constant<int, 4>{}); return true;
        case 8: f(std::integral_constant<int, 8>{}); return true;
        case 16: f(std::integral_constant<int, 16>{}); return true;
        case 32: f(std::integral_constant<int, 32>{}); return true;
        case 64: f(std::integral_constant<int, 64>{}); return true;
        case 128: f(std::integral_constant<int, 128>{}); return true;
        case 256: f(std::integral_constant<int, 256>{}); return true;
        case 512: f(std::integral_constant<int, 512>{}); return true;
        default: return false;
    }
}

// ---------------------------------------------------------------------------
// the engine
// ---------------------------------------------------------------------------
enum ProfKind {
    PK_INV_Z = 0,     // psi_k -> scratch, drift on load, inverse z
    PK_INV_Y,         // inverse y (drift on load in the first half of the step)
    PK_INVX_RHO,      // inverse x + rho + forward x of the Poisson pair
    PK_FWD_X,         // kick on load, forward x, drift on store
    PK_FWD_Y,         // forward y, drift on store
    PK_FWDZ_INVZ,     // forward z, drift, store psi_k, alias sum, inverse z
    PK_W_FWD_Y, PK_POISSON_Z, PK_W_INV_Y, PK_PHI_X,   // Poisson solve on stream pairs
    PK_RHO_X,         // rho from a stored psi (first step, msm_potential)
    PK_PLAIN,         // plain transform passes (msm_fft, first psi_k, downloads)
    PK_COUNT
};

class Engine final : public IEngine {
  public:
    explicit Engine(const msm_config& cfg) : cfg_(cfg) {}
    ~Engine() override { release(); }

    int init(std::string& err) override;
    int set_psi(int s, const void* re, const void* im, std::string& err) override;
    int copy_psi(int dst, int src, std::string& err) override;
    int sample(int s, int scheme, uint64_t seed, std::string& err) override;
    int fft(int s, int dir, std::string& err) override;
    int potential(int first, int count, std::string& err) override;
    int step(std::string& err) override;
    int step_begin(const double* a_now, const double* tau_next, std::string& err) override;
    int step_end(const double* kick_a, const double* new_time, const double* new_tau, const double* snapped_tau,
                 std::string& err) override;
    int sync(std::string& err) override;
    int get_status(msm_stream_status* out, int n, std::string& err) override;
    int get_psi(int s, void* re, void* im, bool kspace, std::string& err) override;
    int get_psi_async(int s, void* re, void* im, int* ticket, std::string& err) override;
    int download_wait(int ticket, std::string& err) override;
    int set_clock(int s, double time, double tau, int current_dumps, std::string& err) override;
    int get_potential(int s, void* phi, std::string& err) override;
    int density_sum(void* dev_out, std::string& err) override;
    int step_density(void* dev_out, std::string& err) override;
    int synth_accumulate(int s, int slot, std::string& err) override;
    int synth_device(int slot, void** dev, long long* count, std::string& err) override;
    int synth_set_count(int slot, long long count, std::string& err) override;
    int synth_fetch(int slot, int field, double* re, double* im, std::string& err) override;
    int synth_qx(int slot, double* re, double* im, std::string& err) override;
    int synth_qx_noalloc(int slot, double* out2, std::string& err) override;
    int synth_release(int slot, std::string& err) override;
    int comm_init(const void* id, int rank, int world, std::string& err) override;
    int mean_density_begin(long long total_streams, std::string& err) override;
    int mean_density(double* host, void** dev, std::string& err) override;
    int synth_allreduce(int slot, std::string& err) override;
    int comm_allreduce_f64(double* vals, int n, int op, std::string& err) override;
    int timer_start(std::string& err) override;
    int timer_stop(float* ms, std::string& err) override;
    int profile(int enable, std::string& err) override;
    int profile_report(int max_kinds, double* total_ms, long long* launches, int* n_kinds, std::string& err) override;
    void derived(double out[8]) const override {
        out[0] = (double)dx_; out[1] = (double)dx_; out[2] = (double)c_.k2max; out[3] = (double)c_.hbar;
        out[4] = cfg_.total_mass / cfg_.particle_mass;
        out[5] = (double)kinetic_dt_; out[6] = (double)comoving_box_; out[7] = two_pass_ ? 2.0 : 0.0;
    }
    int batch() const override { return batch_; }

  private:
    void release();
    int check_stream(int s, std::string& err) const {
        if (s < 0 || s >= ns_) { err = "stream index out of range"; return MSM_ERR_INVALID; }
        return MSM_OK;
    }
    int need_real(std::string& err) const {
        if (fourier_) { err = "the streams are already evolving (state is held in Fourier space)"; return MSM_ERR_STATE; }
        return MSM_OK;
    }
    template <int SGN, int LOAD, int STORE>
    void launch_psi(int ax, const PsiArgs& a, int nblk_y, int kind);
    void fft3_plain(const cplx* src, bool src_local, cplx* dst, bool dst_local, int s0, int n, int sgn, const int* active,
                    bool scaled = true);
    // dims < 3
    PwArgs pw_args(int s0, const int* act) const;
    void lowdim_poisson(const cplx* psi, bool local, real psi_scale, int s0, int nb, bool store_phi, const int* act,
                        bool reduce_max = true);
    void lowdim_step_back(int s0, int nb, const real* nt, const real* ntau, const real* stau);
    PoisArgs pois_args(int s0, int nb, const int* act) const;
#ifdef MSM_TQ
    void tq_rho_from_state(const PoisArgs& p, int nb, bool dens);
#endif
    // W passes after the fused/rho x pass; with_x = false leaves F_y,z-inverted W for a fused consumer
    // (reduce_max = false leaves max|phi| as per-workgroup maxima: the next step's timestep kernel reduces them)
    void poisson_tail(int s0, int nb, bool store_phi, bool track_max, bool with_x = true, const int* act = nullptr,
                      bool phi_for_kick = false, bool reduce_max = true);
    // act = all_active_ solves for finished streams too (standalone msm_potential)
    void poisson_from_psi(const cplx* X, bool local, int s0, int nb, bool store_phi, bool track_max, const int* act);
    // two-pass psi transforms (two_pass_): the kinetic round trip through pass B, psi_k for downloads, and pass A
    // between T and a natural grid.  src/dst are indexed by the launch's local stream number.
    void psi_rt(const cplx* src, cplx* dst, bool alias, int s0, int n, const int* act, int kind);
    void psi_b_fwd(const cplx* src, cplx* dst, int s0, int n);
    void psi_a(int sgn, const cplx* src, bool src_local, cplx* dst, bool dst_local, int s0, int n, const int* act);
    void first_step_init(int s0, int nb);
    // first step: every batch changes representation, also one whose streams have all finished already (a restart)
    void enter_fourier() {
        if (fourier_) return;
        for (int s0 = 0; s0 < ns_; s0 += batch_) first_step_init(s0, std::min(batch_, ns_ - s0));
        fourier_ = true;
    }
    void step_front(int s0, int nb);
    void step_back(int s0, int nb, const real* nt, const real* ntau, const real* stau);
    void publish(bool fresh_max = false);
    template <typename F> void for_batches(F&& f) {
        for (int s0 = 0; s0 < ns_; s0 += batch_) {
            const int nb = std::min(batch_, ns_ - s0);
            bool any = false;
            for (int i = 0; i < nb; ++i) any = any || host_active_[s0 + i];
            if (any) f(s0, nb);
        }
    }

    msm_config cfg_;
    int N_ = 0, ns_ = 0, batch_ = 0, dims_ = 3, ntiles_ = 0, tiles_x_ = 0, tiles_z_ = 0;
    size_t n3_ = 0;
    hipStream_t stream_ = nullptr;
    bool own_stream_ = false;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    cplx *P_ = nullptr;   // [n_streams][n3]: psi before the first step, psi_k afterwards
    cplx *X_ = nullptr;   // [batch][n3]: psi of the batch inside a step
    cplx *W_ = nullptr, *tw_ = nullptr;
    real *PHI_ = nullptr, *ksq_ = nullptr;
    double* partial_ = nullptr;
    int* all_active_ = nullptr;     // [n_streams] of ones
    double* step_dens_ = nullptr;   // caller-owned accumulation target, may be null
    real* d_in_[3] = {nullptr, nullptr, nullptr};  // expanding per-step inputs
    State st_{};
    Consts c_{};
    real dx_ = 0, kinetic_dt_ = 0, comoving_box_ = 0;
    msm_stream_status* h_status_ = nullptr;
    std::vector<int> host_active_;
    bool fourier_ = false;   // P_ holds psi_k (after the first step)
    bool lowdim_ = false;    // dims < 3 or chirp-z lines: whole transforms + pointwise kernels
    int pw_blocks_ = 64;     // blocks per stream of the pointwise kernels (and of their partials)
    // chirp-z line transforms: any even N that the power-of-two line engine does not serve
    bool anyn_ = false;
    int any_M_ = 0, any_lpw_ = 1;
    std::vector<int> mix_radix_;   // non-empty: N is 2-3-5-smooth and takes the mixed-radix line transform instead
    int mix_lpw_ = 1;
    cplx *chirp_ = nullptr, *bhat_ = nullptr, *twm_ = nullptr;
    template <int SGN> void launch_any(int ax, const PsiArgs& a, int nblk_y, int kind);
    bool two_pass_ = false;  // every 3-D transform in two passes (N = 256; 512 in single precision); P_ then holds S = pass A of psi between steps
    bool tq_ = false;        // two_pass_ with the 512^3 tiling (kernels_two_pass_512.inc)
    bool ts_ = false;        // two_pass_ with the 128^3 tiling (kernels_two_pass_128.inc)
    int tp_a_tiles_ = 0, tp_b_blocks_ = 0;   // workgroups per grid of the pass-A / pass-B kernels
    // With S = A[psi] as the state: psi = A^-1[S] / N_A and S' = B^-1[D B[S]] / N_B, where N_A, N_B are the numbers of
    // points the two passes transform over (N_A N_B = N^3; both equal N^{3/2} at 256^3, 16384 and 8192 at 512^3);
    // psi_k = B[S] / N^{3/2} either way.
    real tp_scale_a_ = 0, tp_scale_b_ = 0;
    cplx* stage_ = nullptr;  // one grid: download staging
    bool in_step_ = false;
    // captured static step of small grids
    static constexpr size_t GRAPH_MAX_CELLS = (size_t)1 << 23;
    bool graph_ok_ = true;
    hipGraphExec_t graph_exec_ = nullptr;
    std::vector<char> graph_key_;
    double* graph_dens_ = nullptr;
    int phi_s0_ = -1, phi_nb_ = 0;
    std::vector<void*> allocs_;
    int planes_of(int s, bool kspace, real* stage, std::string& err);
    static constexpr int DL_SLOTS = 4;
    cplx* dl_stage_[DL_SLOTS] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t dl_ready_[DL_SLOTS] = {}, dl_done_[DL_SLOTS] = {};
    hipStream_t copy_stream_ = nullptr;
    int dl_next_ = 0;
    struct SynthSlot { double2* sums = nullptr; long long count = 0; };
    std::vector<SynthSlot> synth_;
    double2* qx_scratch_ = nullptr;   // this device's copy of `scratch`
    int synth_check(int slot, std::string& err) const {
        if (slot < 0 || slot >= (int)synth_.size() || !synth_[slot].sums) { err = "synthesizer slot is not open"; return MSM_ERR_INVALID; }
        return MSM_OK;
    }
    // psi (natural order) and unitary psi_k (reference order) of stream s on the device; valid until the next call
    void materialise(int s, const cplx** psi, const cplx** psik);

    // ---- several GPUs: one context per process and GPU, RCCL between them (bound at run time, comm.hpp)
    const Rccl* rccl_ = nullptr;
    void* comm_ = nullptr;
    int comm_rank_ = 0, comm_world_ = 1;
    hipStream_t comm_stream_ = nullptr;                 // every collective runs here (the per-step density beside the next step)
    hipEvent_t comm_ev_ = nullptr;                      // compute stream -> communication stream
    static constexpr int COMM_SMALL = 64;
    double *comm_small_d_ = nullptr, *comm_small_h_ = nullptr;
    // per-step ensemble density: two sum buffers used alternately, so that the all-reduce of step i overlaps step i + 1
    bool mean_on_ = false;
    long long mean_total_ = 0;
    double* mean_buf_[2] = {nullptr, nullptr};
    double* mean_out_ = nullptr;
    hipEvent_t dens_ready_[2] = {}, comm_done_[2] = {};
    int mean_next_ = 0, mean_last_ = -1;
    int nccl_fail(int rc, const char* what, std::string& err) const {
        err = std::string(what) + ": " + (rccl_ && rccl_->error_string ? rccl_->error_string(rc) : "RCCL error");
        return MSM_ERR_HIP;
    }
    // the density target of the step about to be enqueued / the collective behind the step just enqueued
    int mean_step_open(std::string& err);
    int mean_step_close(std::string& err);

    // optional per-launch timing (HIP events on the launch stream), bucketed by pass kind
    struct ProfRec { int kind; hipEvent_t a, b; };
    std::vector<ProfRec> prof_;
    std::vector<hipEvent_t> ev_pool_;
    bool prof_on_ = false;
    hipEvent_t pool_event() {
        if (!ev_pool_.empty()) { hipEvent_t e = ev_pool_.back(); ev_pool_.pop_back(); return e; }
        hipEvent_t e = nullptr;
        (void)hipEventCreate(&e);
        return e;
    }
    template <typename F> void timed(int kind, F&& launch) {
        if (!prof_on_) { launch(); return; }
        ProfRec r{kind, pool_event(), pool_event()};
        (void)hipEventRecord(r.a, stream_);
        launch();
        (void)hipEventRecord(r.b, stream_);
        prof_.push_back(r);
    }
};

void Engine::release() {
    if (comm_) {
        if (comm_stream_) (void)hipStreamSynchronize(comm_stream_);
        if (stream_) (void)hipStreamSynchronize(stream_);
        (void)rccl_->comm_destroy(comm_);
        comm_ = nullptr;
    }
    for (int i = 0; i < 2; ++i) {
        if (mean_buf_[i]) (void)hipFree(mean_buf_[i]);
        if (dens_ready_[i]) (void)hipEventDestroy(dens_ready_[i]);
        if (comm_done_[i]) (void)hipEventDestroy(comm_done_[i]);
        mean_buf_[i] = nullptr; dens_ready_[i] = comm_done_[i] = nullptr;
    }
    if (mean_out_) { (void)hipFree(mean_out_); mean_out_ = nullptr; }
    if (comm_small_d_) { (void)hipFree(comm_small_d_); comm_small_d_ = nullptr; }
    if (comm_small_h_) { (void)hipHostFree(comm_small_h_); comm_small_h_ = nullptr; }
    if (comm_ev_) { (void)hipEventDestroy(comm_ev_); comm_ev_ = nullptr; }
    if (comm_stream_) { (void)hipStreamDestroy(comm_stream_); comm_stream_ = nullptr; }
    if (copy_stream_) {
        (void)hipStreamSynchronize(copy_stream_);
        for (int i = 0; i < DL_SLOTS; ++i) {
            if (dl_stage_[i]) (void)hipFree(dl_stage_[i]);
            if (dl_ready_[i]) (void)hipEventDestroy(dl_ready_[i]);
            if (dl_done_[i]) (void)hipEventDestroy(dl_done_[i]);
        }
        (void)hipStreamDestroy(copy_stream_);
        copy_stream_ = nullptr;
    }
    for (auto& sl : synth_) if (sl.sums) (void)hipFree(sl.sums);
    synth_.clear();
    if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
    for (void* p : allocs_) (void)hipFree(p);
    allocs_.clear();
    if (h_status_) (void)hipHostFree(h_status_);
    h_status_ = nullptr;
    for (auto& r : prof_) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
    prof_.clear();
    for (hipEvent_t e : ev_pool_) (void)hipEventDestroy(e);
    ev_pool_.clear();
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
    ev0_ = ev1_ = nullptr;
    if (own_stream_ && stream_) (void)hipStreamDestroy(stream_);
    stream_ = nullptr;
}

int Engine::init(std::string& err) {
    N_ = cfg_.size;
    ns_ = cfg_.n_streams;
    dims_ = cfg_.dims;
    if (dims_ < 1 || dims_ > 3) { err = "dims must be 1, 2 or 3"; return MSM_ERR_INVALID; }
    // get_kgrid asserts an even edge (utils/fft.rs:105); powers of two up to 512 have register line transforms
    if (N_ < 2 || N_ > 1024 || (N_ & 1)) { err = "size must be even and in [2, 1024]"; return MSM_ERR_INVALID; }
    anyn_ = (N_ & (N_ - 1)) != 0 || N_ > 512;
    lowdim_ = dims_ < 3 || anyn_;
    pw_blocks_ = dims_ == 3 ? 1024 : 64;
    if (ns_ < 1) { err = "n_streams must be >= 1"; return MSM_ERR_INVALID; }
    if (cfg_.num_data_dumps == 0) { err = "num_data_dumps must be >= 1"; return MSM_ERR_INVALID; }
    n3_ = 1;
    for (int d = 0; d < dims_; ++d) n3_ *= (size_t)N_;   // cells per grid (N^dims)

    HIP_TRY(hipSetDevice(cfg_.device));
    if (cfg_.hip_stream) {
        stream_ = (hipStream_t)cfg_.hip_stream;
    } else {
        HIP_TRY(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
        own_stream_ = true;
    }
    HIP_TRY(hipEventCreate(&ev0_));
    HIP_TRY(hipEventCreate(&ev1_));
    qx_scratch_ = qx_scratch_of(cfg_.device, err);   // cfg_.device is current here
    if (!qx_scratch_) return MSM_ERR_HIP;

    // sub-batch: bounded scratch (X + W + PHI = 2 complex grids per stream), whole Poisson pairs
    batch_ = cfg_.batch_streams > 0 ? cfg_.batch_streams : 32;
    if (batch_ < ns_ && (batch_ & 1)) batch_ += 1;
    if (batch_ > ns_) batch_ = ns_;
    if ((N_ == tp::N || N_ == 512 || N_ == ts::N) && dims_ == 3 && batch_ > tp::MAX_BATCH) batch_ = tp::MAX_BATCH;
    if (batch_ > GENERIC_MAX_BATCH) batch_ = GENERIC_MAX_BATCH;

    // ---- scalars in the reference's arithmetic (SimulationParameters::new, :243-274)
    const real L = (real)cfg_.axis_length;
    const real hbar = (real)cfg_.hbar_;
    real box = L;
    if (cfg_.expanding) {
        const double a0 = 1.0 / (1.0 + cfg_.z0);
        const double comoving = cfg_.axis_length / a0;
        const double hh = 1.022e-4 * cfg_.h;
        comoving_box_ = (real)(std::sqrt(std::sqrt(1.5 * cfg_.omega_matter_now * hh * hh) / (double)hbar) * comoving);
        box = comoving_box_;
    }
    dx_ = box / (real)N_;
    std::vector<real> ksq(N_);
    for (int i = 0; i < N_; ++i) {
        const int ii = i < N_ / 2 ? i : i - N_;
        const real k = (real)ii / ((real)N_ * dx_);   // get_kgrid, utils/fft.rs:100-120
        ksq[i] = k * k;
    }
    const real fourpi2 = std::pow((real)(2.0 * M_PI), (real)2.0);  // utils/fft.rs:154-160
    const real km = ksq[N_ / 2];
    c_.k2max = (dims_ == 3 ? ((km + km) + km) : dims_ == 2 ? (km + km) : km) * fourpi2;   // max of spec_grid, summed in axis order
    c_.cfl = (real)cfg_.cfl;
    c_.box = box;
    c_.hbar = hbar;
    c_.tfinal = (real)cfg_.final_sim_time;
    c_.ndumps = (int)cfg_.num_data_dumps;
    c_.dkpow = std::pow(dx_, (real)dims_);                         // dk = dx, :263, :1281-1285
    c_.alias_thr = (real)cfg_.alias_threshold;
    c_.alias_cut = c_.k2max * (real)cfg_.k2_cutoff;                // :1265
    c_.fourpi2 = fourpi2;
    c_.expanding = cfg_.expanding ? 1 : 0;
    c_.N = N_;
    c_.dims = dims_;
    const double pois_const = 4.0 * M_PI * 4.49e-12;               // common/src/constants.rs:2
    if (cfg_.expanding) {
        const double hh = cfg_.h * 1.022e-4;
        const real f = (real)(cfg_.total_mass * pois_const * std::pow(2.0 / (3.0 * hh * hh * cfg_.omega_matter_now), 0.25));
        c_.mass_factor = f / std::pow(hbar, (real)(dims_ / 2.0));  // :1033-1048
        c_.pois_c = (real)-1.0;
        kinetic_dt_ = c_.cfl * (real)2.0 * box / std::sqrt(c_.k2max);
    } else {
        c_.mass_factor = (real)cfg_.total_mass;
        c_.pois_c = (real)(-pois_const);
        kinetic_dt_ = c_.cfl * (real)2.0 * box / std::sqrt(c_.k2max) / hbar;
    }
    c_.inv_n3 = (real)(1.0 / (double)n3_);
    c_.unit_scale = (real)(1.0 / std::pow((double)N_, dims_ / 2.0));

    if (const char* e = getenv("MSM_GRAPH")) graph_ok_ = atoi(e) != 0;     // 0: plain launches for small grids too
    if (!own_stream_ && stream_ == nullptr) graph_ok_ = false;               // the legacy default stream cannot be captured
    two_pass_ = (N_ == tp::N) && dims_ == 3;
    const char* tp_env = getenv("MSM_TWO_PASS");
    if (tp_env) two_pass_ = two_pass_ && atoi(tp_env) != 0;   // 0: the generic three-pass engine
#ifdef MSM_TQ
    // The single-precision 512^3 tiling (register-resident tiles, kernels_two_pass_512.inc) is the default for that grid;
    // MSM_TWO_PASS=0 selects the three-pass engine (measurements of both: tools/bench_c5.py).
    tq_ = N_ == tq::N && dims_ == 3 && !(tp_env && atoi(tp_env) == 0);
    two_pass_ = two_pass_ || tq_;
    static_assert(tq::MAX_BATCH == tp::MAX_BATCH, "one clamp serves both tilings");
#endif
    ts_ = N_ == ts::N && dims_ == 3 && !(tp_env && atoi(tp_env) == 0);
    two_pass_ = two_pass_ || ts_;
    static_assert(ts::MAX_BATCH == tp::MAX_BATCH, "one clamp serves all tilings");
    tp_a_tiles_ = tq_ ? 4096 : ts_ ? ts::A_TILES : tp::A_TILES;
    tp_b_blocks_ = tq_ ? 4096 : ts_ ? ts::B_BLOCKS : tp::B_BLOCKS;
    tp_scale_a_ = tq_ ? (real)(1.0 / 8192.0) : ts_ ? (real)(1.0 / 2048.0) : c_.unit_scale;
    tp_scale_b_ = tq_ ? (real)(1.0 / 16384.0) : ts_ ? (real)(1.0 / 1024.0) : c_.unit_scale;
    // per-workgroup partials: the x-pass and z-pass tilings may differ
    ntiles_ = pw_blocks_;
    if (!anyn_) dispatch_n(N_, [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        tiles_x_ = (int)(n3_ / N / Geom<N, 0>::LPW);
        tiles_z_ = (int)(n3_ / N / Geom<N, 2>::LPW);
        ntiles_ = std::max(std::max(tiles_x_, tiles_z_), pw_blocks_);
    });

    // ---- device memory
    auto dmalloc = [&](void** p, size_t bytes) -> hipError_t {
        hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) allocs_.push_back(*p);
        return e;
    };
    const size_t grid_bytes = n3_ * sizeof(cplx);
    const int pairs = (batch_ + 1) / 2;
    {
        // The large arrays are checked against the free memory first: an allocation beyond it does not always fail
        // at hipMalloc, it can surface as an out-of-memory error of the first kernel that touches the array.
        const size_t need = grid_bytes * ((size_t)ns_ + 2 * pairs + (lowdim_ ? 2 * pairs : pairs) + 1) +
                            n3_ * sizeof(real) * (size_t)(2 * pairs) + sizeof(double) * (size_t)ns_ * ntiles_;
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        if (need + ((size_t)64 << 20) > free_b) {
            err = "device memory: " + std::to_string(need >> 20) + " MiB needed for " + std::to_string(ns_) + " streams, " +
                  std::to_string(free_b >> 20) + " MiB free";
            return MSM_ERR_NOMEM;
        }
    }
    HIP_TRY(dmalloc((void**)&P_, grid_bytes * ns_));
    HIP_TRY(dmalloc((void**)&X_, grid_bytes * (size_t)(2 * pairs)));
    HIP_TRY(dmalloc((void**)&W_, grid_bytes * (lowdim_ ? 2 * pairs : pairs)));   // dims < 3: one slot per stream
    HIP_TRY(dmalloc((void**)&stage_, grid_bytes));
    HIP_TRY(dmalloc((void**)&PHI_, n3_ * sizeof(real) * (size_t)(2 * pairs)));
    HIP_TRY(dmalloc((void**)&partial_, sizeof(double) * (size_t)ns_ * ntiles_));
    HIP_TRY(dmalloc((void**)&tw_, sizeof(cplx) * N_));
    HIP_TRY(dmalloc((void**)&ksq_, sizeof(real) * N_));
    HIP_TRY(dmalloc((void**)&st_.time, sizeof(real) * ns_));
    HIP_TRY(dmalloc((void**)&st_.tau, sizeof(real) * ns_));
    HIP_TRY(dmalloc((void**)&st_.dt, sizeof(real) * ns_));
    HIP_TRY(dmalloc((void**)&st_.kick, sizeof(real) * ns_));
    HIP_TRY(dmalloc((void**)&st_.a_now, sizeof(real) * ns_));
    HIP_TRY(dmalloc((void**)&st_.tau_next, sizeof(real) * ns_));
    HIP_TRY(dmalloc((void**)&st_.maxphi, sizeof(unsigned long long) * ns_));
    HIP_TRY(dmalloc((void**)&st_.maxphi_step, sizeof(double) * ns_));
    HIP_TRY(dmalloc((void**)&st_.alias_sum, sizeof(double) * ns_));
    HIP_TRY(dmalloc((void**)&st_.active, sizeof(int) * ns_));
    HIP_TRY(dmalloc((void**)&all_active_, sizeof(int) * ns_));
    HIP_TRY(dmalloc((void**)&st_.dump, sizeof(int) * ns_));
    HIP_TRY(dmalloc((void**)&st_.dumps, sizeof(int) * ns_));
    HIP_TRY(dmalloc((void**)&st_.aliased, sizeof(int) * ns_));
    HIP_TRY(dmalloc((void**)&st_.stepped, sizeof(int) * ns_));
    HIP_TRY(dmalloc((void**)&st_.nsteps, sizeof(long long) * ns_));
    HIP_TRY(dmalloc((void**)&st_.drift, sizeof(cplx) * (size_t)ns_ * N_));
    HIP_TRY(dmalloc((void**)&st_.status, sizeof(msm_stream_status) * ns_));
    for (int i = 0; i < 3; ++i) HIP_TRY(dmalloc((void**)&d_in_[i], sizeof(real) * ns_));
    HIP_TRY(hipHostMalloc((void**)&h_status_, sizeof(msm_stream_status) * ns_));

    // ---- tables
    std::vector<cplx> tw(N_);
    for (int p = 0; p < N_; ++p) {
        const long double ang = -2.0L * 3.141592653589793238462643383279502884L * (long double)p / (long double)N_;
        tw[p].x = (real)cosl(ang);
        tw[p].y = (real)sinl(ang);
    }
    HIP_TRY(hipMemcpy(tw_, tw.data(), sizeof(cplx) * N_, hipMemcpyHostToDevice));
    if (anyn_ && getenv("MSM_CHIRPZ") == nullptr) {
        // 2-3-5-smooth edges: radix list of the mixed-radix line transform (radix 4 first, then 2, 3, 5)
        int m = N_;
        std::vector<int> rad;
        while (m % 4 == 0) { rad.push_back(4); m /= 4; }
        while (m % 2 == 0) { rad.push_back(2); m /= 2; }
        while (m % 3 == 0) { rad.push_back(3); m /= 3; }
        while (m % 5 == 0) { rad.push_back(5); m /= 5; }
        if (m == 1 && rad.size() <= 12) {
            mix_radix_ = rad;
            mix_lpw_ = 1;
            for (int c : {8, 4, 2}) if (N_ % c == 0 && (size_t)2 * c * N_ * sizeof(cplx) <= 65536) { mix_lpw_ = c; break; }
        }
    }
    if (anyn_ && mix_radix_.empty()) {
        // chirp-z tables in extended precision: c[n] = exp(-i pi n^2 / N) with n^2 reduced mod 2N exactly,
        // the length-M spectrum of the wrapped conj(c) (bit-reversed, 1/M folded in), M-th roots of unity
        typedef std::complex<long double> lc;
        const long double pi = 3.141592653589793238462643383279502884L;
        int M = 1;
        while (M < 2 * N_ - 1) M <<= 1;
        any_M_ = M;
        any_lpw_ = 1;
        for (int c : {8, 4, 2}) if (N_ % c == 0 && (size_t)c * M * sizeof(cplx) <= 65536) { any_lpw_ = c; break; }
        std::vector<lc> ch(N_), b(M, lc(0, 0)), w(M / 2);
        for (int n = 0; n < N_; ++n) {
            const long long r = ((long long)n * n) % (2LL * N_);
            const long double ang = -pi * (long double)r / (long double)N_;
            ch[n] = lc(cosl(ang), sinl(ang));
        }
        b[0] = lc(1, 0);
        for (int n = 1; n < N_; ++n) b[n] = b[M - n] = std::conj(ch[n]);
        for (int k = 0; k < M / 2; ++k) {
            const long double ang = -2.0L * pi * (long double)k / (long double)M;
            w[k] = lc(cosl(ang), sinl(ang));
        }
        for (int half = M / 2, ts = 1; half >= 1; half >>= 1, ts <<= 1)      // decimation in frequency, output bit-reversed
            for (int g = 0; g < M; g += 2 * half)
                for (int j = 0; j < half; ++j) {
                    const lc u = b[g + j], v = b[g + j + half];
                    b[g + j] = u + v;
                    b[g + j + half] = (u - v) * w[(size_t)j * ts];
                }
        std::vector<cplx> chr(N_), bh(M), twm(M / 2);
        for (int n = 0; n < N_; ++n) { chr[n].x = (real)ch[n].real(); chr[n].y = (real)ch[n].imag(); }
        for (int i = 0; i < M; ++i) { bh[i].x = (real)(b[i].real() / M); bh[i].y = (real)(b[i].imag() / M); }
        for (int k = 0; k < M / 2; ++k) { twm[k].x = (real)w[k].real(); twm[k].y = (real)w[k].imag(); }
        HIP_TRY(dmalloc((void**)&chirp_, sizeof(cplx) * N_));
        HIP_TRY(dmalloc((void**)&bhat_, sizeof(cplx) * M));
        HIP_TRY(dmalloc((void**)&twm_, sizeof(cplx) * (M / 2)));
        HIP_TRY(hipMemcpy(chirp_, chr.data(), sizeof(cplx) * N_, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(bhat_, bh.data(), sizeof(cplx) * M, hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(twm_, twm.data(), sizeof(cplx) * (M / 2), hipMemcpyHostToDevice));
    }
    HIP_TRY(hipMemcpy(ksq_, ksq.data(), sizeof(real) * N_, hipMemcpyHostToDevice));

    // ---- per-stream state
    std::vector<real> t0(ns_, (real)cfg_.time), zr(ns_, (real)0);
    std::vector<int> ones(ns_, 1), zi(ns_, 0);
    std::vector<long long> zl(ns_, 0);
    std::vector<cplx> one_c((size_t)ns_ * N_);
    for (auto& v : one_c) { v.x = (real)1; v.y = (real)0; }
    HIP_TRY(hipMemcpy(st_.time, t0.data(), sizeof(real) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.tau, zr.data(), sizeof(real) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.dt, zr.data(), sizeof(real) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.kick, zr.data(), sizeof(real) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.a_now, zr.data(), sizeof(real) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.tau_next, zr.data(), sizeof(real) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(st_.maxphi, 0, sizeof(unsigned long long) * ns_));
    HIP_TRY(hipMemset(st_.maxphi_step, 0, sizeof(double) * ns_));
    HIP_TRY(hipMemset(st_.alias_sum, 0, sizeof(double) * ns_));
    HIP_TRY(hipMemcpy(st_.active, ones.data(), sizeof(int) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(all_active_, ones.data(), sizeof(int) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.dump, zi.data(), sizeof(int) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.dumps, zi.data(), sizeof(int) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.aliased, zi.data(), sizeof(int) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.stepped, zi.data(), sizeof(int) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.nsteps, zl.data(), sizeof(long long) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.drift, one_c.data(), sizeof(cplx) * (size_t)ns_ * N_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(P_, 0, grid_bytes * ns_));
    if (const char* e = getenv("MSM_POISON"); e && atoi(e) != 0) {
        // diagnostic: the work arrays start as NaN bit patterns, so a read of memory that no kernel has written shows
        // in the results instead of depending on what the allocation held before
        HIP_TRY(hipMemset(X_, 0xFF, grid_bytes * (size_t)(2 * pairs)));
        HIP_TRY(hipMemset(W_, 0xFF, grid_bytes * (lowdim_ ? 2 * pairs : pairs)));
        HIP_TRY(hipMemset(stage_, 0xFF, grid_bytes));
        HIP_TRY(hipMemset(PHI_, 0xFF, n3_ * sizeof(real) * (size_t)(2 * pairs)));
        HIP_TRY(hipMemset(partial_, 0xFF, sizeof(double) * (size_t)ns_ * ntiles_));
    }
    host_active_.assign(ns_, 1);
    publish();
    HIP_TRY(hipStreamSynchronize(stream_));
    return MSM_OK;
}

void Engine::publish(bool fresh_max) {
    k_publish<<<(ns_ + 63) / 64, 64, 0, stream_>>>(st_, c_, ns_, fresh_max ? 1 : 0);
    (void)hipMemcpyAsync(h_status_, st_.status, sizeof(msm_stream_status) * ns_, hipMemcpyDeviceToHost, stream_);
}

template <int SGN>
void Engine::launch_any(int ax, const PsiArgs& p, int nblk_y, int kind) {
    if (!mix_radix_.empty()) {
        MixArgs m{};
        m.src = p.src; m.dst = p.dst; m.n3 = p.n3; m.active = p.active; m.s0 = p.s0; m.src_local = p.src_local;
        m.dst_local = p.dst_local; m.N = N_; m.tw = tw_; m.scale = p.scale; m.nst = (int)mix_radix_.size();
        for (int i = 0; i < m.nst; ++i) m.radix[i] = mix_radix_[i];
        m.stride = 1;
        for (int i = 0; i < ax; ++i) m.stride *= (size_t)N_;
        const size_t lines = n3_ / (size_t)N_;
        m.lpw = (int)std::min<size_t>((size_t)mix_lpw_, lines);
        const dim3 grid((unsigned)(lines / m.lpw), (unsigned)nblk_y);
        const size_t lds = (size_t)2 * m.lpw * N_ * sizeof(cplx);
        timed(kind, [&] { k_line_mixed<SGN><<<grid, dim3(256), lds, stream_>>>(m); });
        return;
    }
    AnyArgs a{};
    a.src = p.src; a.dst = p.dst; a.n3 = p.n3; a.active = p.active; a.s0 = p.s0; a.src_local = p.src_local;
    a.dst_local = p.dst_local; a.N = N_; a.M = any_M_; a.chirp = chirp_; a.bhat = bhat_; a.twm = twm_; a.scale = p.scale;
    a.stride = 1;
    for (int i = 0; i < ax; ++i) a.stride *= (size_t)N_;
    const size_t lines = n3_ / (size_t)N_;
    a.lpw = (int)std::min<size_t>((size_t)any_lpw_, lines);        // a 1-D grid is one line
    const dim3 grid((unsigned)(lines / a.lpw), (unsigned)nblk_y);
    const size_t lds = (size_t)a.lpw * any_M_ * sizeof(cplx);
    timed(kind, [&] { k_line_any<SGN><<<grid, dim3(256), lds, stream_>>>(a); });
}

template <int SGN, int LOAD, int STORE>
void Engine::launch_psi(int ax, const PsiArgs& a0, int nblk_y, int kind) {
    if (anyn_) { launch_any<SGN>(ax, a0, nblk_y, kind); return; }   // only plain passes exist on this path
    PsiArgs a = a0;
    const size_t lines = n3_ / (size_t)N_;
    a.nlines = (int)std::min<size_t>(lines, (size_t)1 << 30);
    timed(kind, [&] {
        dispatch_n(N_, [&](auto nc) {
            constexpr int N = decltype(nc)::value;
            using G0 = Geom<N, 0>;
            using GS = Geom<N, 1>;
            const dim3 grid0((unsigned)((lines + G0::LPW - 1) / G0::LPW), (unsigned)nblk_y), gridS((unsigned)(lines / GS::LPW), (unsigned)nblk_y);
            if (ax == 0) k_psi_pass<N, SGN, 0, LOAD, STORE><<<grid0, dim3(G0::THREADS), 0, stream_>>>(a);
            else if (ax == 1) k_psi_pass<N, SGN, 1, LOAD, STORE><<<gridS, dim3(GS::THREADS), 0, stream_>>>(a);
            else k_psi_pass<N, SGN, 2, LOAD, STORE><<<gridS, dim3(GS::THREADS), 0, stream_>>>(a);
        });
    });
}

// unitary transform over all dims_ axes without pointwise work: first pass src -> dst, rest in place on dst
void Engine::fft3_plain(const cplx* src, bool src_local, cplx* dst, bool dst_local, int s0, int n, int sgn,
                        const int* active, bool scaled) {
    PsiArgs a{};
    a.n3 = n3_; a.active = active; a.s0 = s0; a.pair_nb = 0; a.tw = tw_; a.scale = (real)1;
    a.src = src; a.dst = dst; a.src_local = src_local; a.dst_local = dst_local;
    for (int i = 0; i < dims_; ++i) {
        const int ax = sgn < 0 ? i : dims_ - 1 - i;
        if (i == dims_ - 1 && scaled) a.scale = c_.unit_scale;
        if (sgn < 0) launch_psi<-1, 0, 0>(ax, a, n, PK_PLAIN);
        else launch_psi<+1, 0, 0>(ax, a, n, PK_PLAIN);
        a.src = dst; a.src_local = dst_local;
    }
}

PoisArgs Engine::pois_args(int s0, int nb, const int* act) const {
    PoisArgs p{};
    p.X = X_; p.Xout = X_; p.x_local = 1; p.W = W_; p.PHI = PHI_; p.n3 = n3_; p.active = act;
    p.s0 = s0; p.nb = nb; p.tw = tw_; p.drift = st_.drift; p.ksq = ksq_; p.mass_factor = c_.mass_factor;
    p.pois_c = c_.pois_c; p.fourpi2 = c_.fourpi2; p.inv_n3 = c_.inv_n3; p.scale = c_.unit_scale;
    p.partial = partial_; p.ntiles = ntiles_; p.dens = nullptr;
    return p;
}

#ifdef MSM_TQ
// 512^3 tiling: the batch's states -> pass A^-1 -> |psi|^2 -> pass A of rho_a + i rho_b into W.  The first stream of
// every pair leaves |psi_a|^2 as a real grid (in the potential array), the second stream's workgroups read it and pack
// the pair in registers; an unpaired last stream goes through k5_rho_fwd.  With `dens` every stream's |psi|^2 is kept
// as a grid and summed into the ensemble density of the streams in p.active.
// Launch rows of the 512^3 pass-A kernels that walk their tile through several streams (k5_inv_rho, k5_phi_out): two rows
// share the streams of a batch, so that a workgroup has a next tile to request while it finishes the current one.
static unsigned tq_rows(int count) { return (unsigned)std::min(count, 2); }
void Engine::tq_rho_from_state(const PoisArgs& p, int nb, bool dens) {
    const dim3 block(tq::THREADS);
    k5_inv_rho<2><<<dim3(tq::A_TILES, tq_rows((nb + 1) / 2)), block, 0, stream_>>>(p);
    if (nb >= 2) {
        const dim3 grid(tq::A_TILES, (unsigned)(nb / 2));
        if (dens) k5_inv_rho_pack<1><<<grid, block, 0, stream_>>>(p);
        else k5_inv_rho_pack<0><<<grid, block, 0, stream_>>>(p);
    }
    if (dens) k5_dens_from_rho<<<2048, 256, 0, stream_>>>(PHI_, n3_, nb, p.active, p.s0, p.dens);
    if (nb & 1) k5_rho_fwd<1><<<dim3(tq::A_TILES, 1), block, 0, stream_>>>(p);
}
#endif

// y, z (fused with the Green's multiply), y^-1, x^-1 of the Poisson solve; W already holds F_x[rho]
// (two-pass layout: W holds T and one pass-B kernel replaces the three middle passes)
void Engine::poisson_tail(int s0, int nb, bool store_phi, bool track_max, bool with_x, const int* act, bool phi_for_kick,
                          bool reduce_max) {
    if (!act) act = st_.active;
    const int pairs = (nb + 1) / 2;
    PoisArgs p = pois_args(s0, nb, act);
    if (two_pass_) {
#ifdef MSM_TQ
        if (tq_) {
            timed(PK_POISSON_Z, [&] { k5_poisson_b<<<dim3(tq::B_BLOCKS, (unsigned)pairs), dim3(tq::THREADS), 0, stream_>>>(p); });
            const dim3 grid(tq::A_TILES, tq_rows(pairs)), block(tq::THREADS);
            if (with_x) timed(PK_PHI_X, [&] {
                if (store_phi && phi_for_kick && track_max) k5_phi_out<2, 1><<<grid, block, 0, stream_>>>(p);
                else if (store_phi && phi_for_kick) k5_phi_out<2, 0><<<grid, block, 0, stream_>>>(p);
                else if (store_phi && track_max) k5_phi_out<1, 1><<<grid, block, 0, stream_>>>(p);
                else if (store_phi) k5_phi_out<1, 0><<<grid, block, 0, stream_>>>(p);
                else k5_phi_out<0, 1><<<grid, block, 0, stream_>>>(p);
            });
        } else
#endif
        if (ts_) {
            timed(PK_POISSON_Z, [&] { ks_poisson_b<<<dim3(ts::B_BLOCKS, (unsigned)pairs), dim3(ts::THREADS), 0, stream_>>>(p); });
            const dim3 grid(ts::A_TILES, (unsigned)pairs), block(ts::THREADS);
            if (with_x) timed(PK_PHI_X, [&] {
                if (store_phi && track_max) ks_phi_out<1, 1><<<grid, block, 0, stream_>>>(p);
                else if (store_phi) ks_phi_out<1, 0><<<grid, block, 0, stream_>>>(p);
                else ks_phi_out<0, 1><<<grid, block, 0, stream_>>>(p);
            });
        } else {
            timed(PK_POISSON_Z, [&] { k_poisson_b<<<dim3(tp::B_BLOCKS, (unsigned)pairs), dim3(tp::B_THREADS), 0, stream_>>>(p); });
            const dim3 grid(tp::A_TILES, (unsigned)pairs), block(tp::A_THREADS);
            if (with_x) timed(PK_PHI_X, [&] {
                if (store_phi && track_max) k_phi_out2<1, 1><<<grid, block, 0, stream_>>>(p);
                else if (store_phi) k_phi_out2<1, 0><<<grid, block, 0, stream_>>>(p);
                else k_phi_max3<<<dim3(tp::A_TILES), block, 0, stream_>>>(p);
            });
        }
        if (with_x && track_max && reduce_max) k_reduce_max<<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, tp_a_tiles_, s0, act);
        if (with_x && store_phi) { phi_s0_ = s0; phi_nb_ = phi_for_kick ? 0 : nb; }   // (tile order is not for downloads)
        return;
    }
    PsiArgs a{};
    a.src = W_; a.dst = W_; a.n3 = n3_; a.active = act; a.s0 = s0; a.pair_nb = nb; a.tw = tw_; a.scale = (real)1;
    dispatch_n(N_, [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        using G0 = Geom<N, 0>;
        using GS = Geom<N, 1>;
        const dim3 grid0((unsigned)(n3_ / N / G0::LPW), (unsigned)pairs), gridS((unsigned)(n3_ / N / GS::LPW), (unsigned)pairs);
        const dim3 block0(G0::THREADS), blockS(GS::THREADS);
        timed(PK_W_FWD_Y, [&] { k_psi_pass<N, -1, 1, 0, 0><<<gridS, blockS, 0, stream_>>>(a); });
        timed(PK_POISSON_Z, [&] { k_poisson_mid<N><<<gridS, blockS, 0, stream_>>>(p); });
        timed(PK_W_INV_Y, [&] { k_psi_pass<N, +1, 1, 0, 0><<<gridS, blockS, 0, stream_>>>(a); });
        if (with_x) timed(PK_PHI_X, [&] {
            if (store_phi && track_max) k_phi_out<N, 1, 1><<<grid0, block0, 0, stream_>>>(p);
            else if (store_phi) k_phi_out<N, 1, 0><<<grid0, block0, 0, stream_>>>(p);
            else k_phi_out<N, 0, 1><<<grid0, block0, 0, stream_>>>(p);
        });
    });
    if (with_x && track_max && reduce_max) k_reduce_max<<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, tiles_x_, s0, act);
    if (with_x && store_phi) { phi_s0_ = s0; phi_nb_ = nb; }
}

// whole Poisson solve from a psi that sits in memory
void Engine::poisson_from_psi(const cplx* X, bool local, int s0, int nb, bool store_phi, bool track_max, const int* act) {
    PoisArgs p = pois_args(s0, nb, act);
    p.X = X; p.x_local = local ? 1 : 0;
    if (two_pass_) {
#ifdef MSM_TQ
        if (tq_) timed(PK_RHO_X, [&] { k5_rho_pass<<<dim3(tq::A_TILES, (unsigned)((nb + 1) / 2)), dim3(tq::THREADS), 0, stream_>>>(p); });
        else
#endif
        if (ts_) timed(PK_RHO_X, [&] { ks_rho_pass<<<dim3(ts::A_TILES, (unsigned)((nb + 1) / 2)), dim3(ts::THREADS), 0, stream_>>>(p); });
        else timed(PK_RHO_X, [&] { k_rho_pass2<<<dim3(tp::A_TILES, (unsigned)((nb + 1) / 2)), dim3(tp::A_THREADS), 0, stream_>>>(p); });
    } else dispatch_n(N_, [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        using G0 = Geom<N, 0>;
        const dim3 grid0((unsigned)(n3_ / N / G0::LPW), (unsigned)((nb + 1) / 2));
        timed(PK_RHO_X, [&] { k_rho_pass<N><<<grid0, dim3(G0::THREADS), 0, stream_>>>(p); });
    });
    poisson_tail(s0, nb, store_phi, track_max, true, act);
}

void Engine::psi_rt(const cplx* src, cplx* dst, bool alias, int s0, int n, const int* act, int kind) {
    PsiBArgs b{};
    b.src = src; b.dst = dst; b.n3 = n3_; b.active = act; b.s0 = s0; b.tw = tw_; b.drift = st_.drift; b.ksq = ksq_;
    b.scale = tp_scale_b_; b.alias_cut = c_.alias_cut; b.fourpi2 = c_.fourpi2; b.partial = partial_; b.ntiles = ntiles_;
    // the aliased mass is that of the unitary psi_k = B[S] D / N^{3/2}; the kernel sums |B[S] D / N_B|^2
    b.alias_scale = (double)(c_.unit_scale / tp_scale_b_) * (double)(c_.unit_scale / tp_scale_b_);
#ifdef MSM_TQ
    if (tq_) {
        const dim3 grid(tq::B_BLOCKS, (unsigned)n), block(tq::THREADS);
        timed(kind, [&] {
            if (alias) k5_psi_rt<1><<<grid, block, 0, stream_>>>(b);
            else k5_psi_rt<0><<<grid, block, 0, stream_>>>(b);
        });
        return;
    }
#endif
    if (ts_) {
        const dim3 grid(ts::B_BLOCKS, (unsigned)n), block(ts::THREADS);
        timed(kind, [&] {
            if (alias) ks_psi_rt<1><<<grid, block, 0, stream_>>>(b);
            else ks_psi_rt<0><<<grid, block, 0, stream_>>>(b);
        });
        return;
    }
    const dim3 grid(tp::B_BLOCKS, (unsigned)n), block(tp::B_THREADS);
    timed(kind, [&] {
        if (alias) k_psi_rt<1><<<grid, block, 0, stream_>>>(b);
        else k_psi_rt<0><<<grid, block, 0, stream_>>>(b);
    });
}

void Engine::psi_b_fwd(const cplx* src, cplx* dst, int s0, int n) {
    PsiBArgs b{};
    b.src = src; b.dst = dst; b.n3 = n3_; b.active = nullptr; b.s0 = s0; b.tw = tw_; b.scale = c_.unit_scale;
#ifdef MSM_TQ
    if (tq_) { timed(PK_PLAIN, [&] { k5_psi_b_fwd<<<dim3(tq::B_BLOCKS, (unsigned)n), dim3(tq::THREADS), 0, stream_>>>(b); }); return; }
#endif
    if (ts_) { timed(PK_PLAIN, [&] { ks_psi_b_fwd<<<dim3(ts::B_BLOCKS, (unsigned)n), dim3(ts::THREADS), 0, stream_>>>(b); }); return; }
    timed(PK_PLAIN, [&] { k_psi_b_fwd<<<dim3(tp::B_BLOCKS, (unsigned)n), dim3(tp::B_THREADS), 0, stream_>>>(b); });
}

void Engine::psi_a(int sgn, const cplx* src, bool src_local, cplx* dst, bool dst_local, int s0, int n, const int* act) {
    PsiArgs a{};
    a.src = src; a.dst = dst; a.n3 = n3_; a.active = act; a.s0 = s0; a.src_local = src_local; a.dst_local = dst_local;
    a.tw = tw_; a.scale = tp_scale_a_;
#ifdef MSM_TQ
    if (tq_) {
        const dim3 grid(tq::A_TILES, (unsigned)n), block(tq::THREADS);
        timed(PK_PLAIN, [&] {
            if (sgn < 0) k5_psi_a<-1><<<grid, block, 0, stream_>>>(a);
            else k5_psi_a<+1><<<grid, block, 0, stream_>>>(a);
        });
        return;
    }
#endif
    if (ts_) {
        const dim3 grid(ts::A_TILES, (unsigned)n), block(ts::THREADS);
        timed(PK_PLAIN, [&] {
            if (sgn < 0) ks_psi_a<-1><<<grid, block, 0, stream_>>>(a);
            else ks_psi_a<+1><<<grid, block, 0, stream_>>>(a);
        });
        return;
    }
    {
        const dim3 grid(tp::A_TILES, (unsigned)n), block(tp::A_THREADS);
        timed(PK_PLAIN, [&] {
            if (sgn < 0) k_psi_a<-1><<<grid, block, 0, stream_>>>(a);
            else k_psi_a<+1><<<grid, block, 0, stream_>>>(a);
        });
    }
}

PwArgs Engine::pw_args(int s0, const int* act) const {
    PwArgs a{};
    a.W = W_; a.PHI = PHI_; a.n = n3_; a.active = act; a.s0 = s0; a.N = N_; a.dims = dims_; a.drift = st_.drift; a.ksq = ksq_;
    a.kick = st_.kick; a.scale = (real)1; a.mass_factor = c_.mass_factor; a.pois_c = c_.pois_c; a.fourpi2 = c_.fourpi2;
    a.inv_n = c_.inv_n3; a.alias_cut = c_.alias_cut; a.partial = partial_; a.ntiles = ntiles_;
    return a;
}

// dims < 3: calculate_potential (simulation_object.rs:1066-1110) as the reference spells it: density, transform,
// Green's multiply, transform back; max|phi| per stream; W (one slot per stream) is left holding phi in its real part.
void Engine::lowdim_poisson(const cplx* psi, bool local, real psi_scale, int s0, int nb, bool store_phi, const int* act,
                            bool reduce_max) {
    PwArgs a = pw_args(s0, act);
    a.src = psi; a.src_local = local ? 1 : 0; a.scale = psi_scale;
    const dim3 grid(pw_blocks_, (unsigned)nb), block(256);
    k_pw_rho<<<grid, block, 0, stream_>>>(a);
    fft3_plain(W_, true, W_, true, s0, nb, -1, act, false);
    k_pw_green<<<grid, block, 0, stream_>>>(a);
    fft3_plain(W_, true, W_, true, s0, nb, +1, act, false);
    if (store_phi) k_pw_phi<1><<<grid, block, 0, stream_>>>(a);
    else k_pw_phi<0><<<grid, block, 0, stream_>>>(a);
    if (reduce_max) k_reduce_max<<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, pw_blocks_, s0, act);
    if (store_phi) { phi_s0_ = s0; phi_nb_ = nb; }
}

// dims < 3: the rest of update (:504-635) with whole transforms and one kernel per pointwise operator
void Engine::lowdim_step_back(int s0, int nb, const real* nt, const real* ntau, const real* stau) {
    const dim3 grid(pw_blocks_, (unsigned)nb), block(256);
    PwArgs a = pw_args(s0, st_.active);
    // psi = F^-1[ D psi_k ]
    a.src = P_; a.src_local = 0; a.dst = X_; a.dst_local = 1;
    k_pw_drift<0><<<grid, block, 0, stream_>>>(a);
    fft3_plain(X_, true, X_, true, s0, nb, +1, st_.active);
    // phi at the half-drifted psi, kick
    lowdim_poisson(X_, true, (real)1, s0, nb, false, st_.active, false);   // (its max|phi| is not used)
    a.dst = X_; a.scale = (real)1;
    k_pw_kick<<<grid, block, 0, stream_>>>(a);
    // psi_k = D F[psi], aliased mass
    fft3_plain(X_, true, X_, true, s0, nb, -1, st_.active);
    a.src = X_; a.src_local = 1; a.dst = P_; a.dst_local = 0; a.scale = (real)1;
    k_pw_drift<1><<<grid, block, 0, stream_>>>(a);
    k_post_step<<<nb, 256, 0, stream_>>>(st_, c_, s0, partial_, ntiles_, pw_blocks_, nt, ntau, stau);
    // potential of psi(t + dt) for the next timestep (:497); the tail also runs for a stream that has just reached
    // the final time, so that its last density is part of msm_step_density's product
    fft3_plain(P_, false, X_, true, s0, nb, +1, st_.stepped);
    if (step_dens_)
        k_density_add_active<<<(unsigned)std::min<size_t>((n3_ + 255) / 256, 2048), 256, 0, stream_>>>(X_, n3_, nb, st_.stepped, s0, step_dens_);
    lowdim_poisson(X_, true, (real)1, s0, nb, false, st_.stepped, false);   // max|phi| is reduced by the next k_front
}

// Before the first step P_ holds psi: size the first timestep from it (simulation_object.rs:497)
// and move the state to Fourier space (psi_k = F psi, :477-479).
void Engine::first_step_init(int s0, int nb) {
    if (lowdim_) {
        lowdim_poisson(P_, false, (real)1, s0, nb, false, st_.active);
        fft3_plain(P_, false, P_, false, s0, nb, -1, nullptr);   // finished streams too: downloads read the new form
        return;
    }
    poisson_from_psi(P_, false, s0, nb, false, true, st_.active);
    if (two_pass_) {
        // the state becomes S = pass A of psi, for every stream of the batch (finished ones too: downloads read S)
        psi_a(-1, P_, false, X_, true, s0, nb, nullptr);
        (void)hipMemcpyAsync(P_ + (size_t)s0 * n3_, X_, sizeof(cplx) * n3_ * (size_t)nb, hipMemcpyDeviceToDevice, stream_);
    } else {
        fft3_plain(P_, false, P_, false, s0, nb, -1, nullptr);
    }
}

// timestep (simulation_object.rs:500) from the per-workgroup maxima of |phi| that the previous step's tail (or the
// first step's set-up, or msm_potential) left; in a static box the kinetic drift table follows in the same launch
void Engine::step_front(int s0, int nb) {
    const int nvalid = lowdim_ ? pw_blocks_ : two_pass_ ? tp_a_tiles_ : tiles_x_;
    if (cfg_.expanding) k_front<0><<<nb, 256, 0, stream_>>>(st_, c_, ksq_, partial_, ntiles_, nvalid, s0);
    else k_front<1><<<nb, 256, 0, stream_>>>(st_, c_, ksq_, partial_, ntiles_, nvalid, s0);
}

// drift/2, F^-1, potential, kick, F, drift/2, F^-1, bookkeeping, potential for the next dt (:504-635, :497)
void Engine::step_back(int s0, int nb, const real* nt, const real* ntau, const real* stau) {
    if (cfg_.expanding) k_drift_table<<<dim3((N_ + 63) / 64, nb), 64, 0, stream_>>>(st_, c_, ksq_, s0);   // else: k_front
    PsiArgs a{};
    a.n3 = n3_; a.active = st_.active; a.s0 = s0; a.pair_nb = 0; a.tw = tw_; a.drift = st_.drift;
    a.phi = PHI_; a.kick = st_.kick; a.scale = (real)1;
    PoisArgs p = pois_args(s0, nb, st_.active);

    if (lowdim_) { lowdim_step_back(s0, nb, nt, ntau, stau); return; }
    if (two_pass_) {
        const dim3 gridA(tp_a_tiles_), gridAp(tp_a_tiles_, (unsigned)((nb + 1) / 2)), blockA(512);
        static_assert(tp::A_THREADS == 512 && ts::THREADS == 512, "all tilings launch their A kernels with 512 threads");
        // one launch of the fused inverse-x / density / Poisson-forward pass, and of the kick pass, in either tiling
        auto invx_rho = [&](bool dens) {
            timed(PK_INVX_RHO, [&] {
#ifdef MSM_TQ
                if (tq_) { tq_rho_from_state(p, nb, dens); return; }
#endif
                if (ts_) {
                    if (dens) ks_invx_rho<1><<<gridA, blockA, 0, stream_>>>(p);
                    else ks_invx_rho<0><<<gridA, blockA, 0, stream_>>>(p);
                    return;
                }
                if (dens) k_invx_rho3<1><<<gridA, blockA, 0, stream_>>>(p);
                else k_invx_rho3<0><<<gridA, blockA, 0, stream_>>>(p);
            });
        };
        auto kick = [&] {
            timed(PK_FWD_X, [&] {
#ifdef MSM_TQ
                if (tq_) { k5_kick<<<dim3(tq::A_TILES, (unsigned)nb), blockA, 0, stream_>>>(p, st_.kick); return; }
#endif
                if (ts_) { ks_kick<<<gridAp, blockA, 0, stream_>>>(p, st_.kick); return; }
                k_kick3<<<gridAp, blockA, 0, stream_>>>(p, st_.kick);
            });
        };
        // The whole step runs in place on the batch's states S (contiguous in P_): every kernel reads a tile or a
        // run completely before it stores it, and tiles and runs are disjoint between workgroups.
        cplx* S = P_ + (size_t)s0 * n3_;
        p.X = S; p.Xout = S;
        p.scale = tp_scale_a_;
        // psi = F^-1[ D psi_k ]: round trip of S through pass B with the whole drift factor in between, then
        // pass A^-1 fused with rho and the Poisson pass A
        psi_rt(S, S, false, s0, nb, st_.active, PK_INV_Z);
        invx_rho(false);
        // (the 512^3 tiling hands phi to the kick as a real grid: its Poisson solve ends with the A^-1 pass)
        poisson_tail(s0, nb, tq_, false, tq_, nullptr, tq_);
        // kick between the two A passes; the second round trip applies the second drift, sums the aliased mass
        // and leaves the new state
        kick();
        psi_rt(S, S, true, s0, nb, st_.active, PK_FWDZ_INVZ);
        k_post_step<<<nb, 256, 0, stream_>>>(st_, c_, s0, partial_, ntiles_, tp_b_blocks_, nt, ntau, stau);
        // potential of psi(t + dt) straight from the state, for every stream that took the step (one that has
        // just reached the final time included: its last density belongs to msm_step_density's product)
        p.dens = step_dens_;
        p.active = st_.stepped;
        invx_rho(step_dens_ != nullptr);
        poisson_tail(s0, nb, false, true, true, st_.stepped, false, false);   // max|phi| is reduced by the next k_front
        if (tq_) phi_nb_ = 0;   // the rho grids went through the potential array
        return;
    }

    // psi = F^-1[ D psi_k ]  (z, y plain passes; x fused with rho and the Poisson forward x pass)
    a.src = P_; a.src_local = 0; a.dst = X_; a.dst_local = 1;
    launch_psi<+1, 1, 0>(2, a, nb, PK_INV_Z);
    a.src = X_; a.src_local = 1;
    launch_psi<+1, 1, 0>(1, a, nb, PK_INV_Y);
    dispatch_n(N_, [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        using G0 = Geom<N, 0>;
        timed(PK_INVX_RHO, [&] {
            k_invx_rho<N, 1, 0, 0><<<dim3((unsigned)(n3_ / N / G0::LPW)), dim3(G0::THREADS), 0, stream_>>>(p);
        });
    });
    // phi at the half-drifted psi, left as F_x[phi_a + i phi_b] in W
    poisson_tail(s0, nb, false, false, false);
    // psi_k = D F[ kick psi ], then on into psi = F^-1[ psi_k ]
    dispatch_n(N_, [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        using G0 = Geom<N, 0>;
        timed(PK_FWD_X, [&] {
            k_kick_fwdx<N><<<dim3((unsigned)(n3_ / N / G0::LPW), (unsigned)((nb + 1) / 2)), dim3(G0::THREADS), 0, stream_>>>(p, st_.kick);
        });
    });
    launch_psi<-1, 0, 1>(1, a, nb, PK_FWD_Y);
    FzArgs f{};
    f.X = X_; f.Xout = X_; f.P = P_; f.n3 = n3_; f.active = st_.active; f.s0 = s0; f.tw = tw_; f.drift = st_.drift;
    f.scale = c_.unit_scale; f.ksq = ksq_; f.alias_cut = c_.alias_cut; f.fourpi2 = c_.fourpi2;
    f.partial = partial_; f.ntiles = ntiles_;
    dispatch_n(N_, [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        using GS = Geom<N, 2>;
        timed(PK_FWDZ_INVZ, [&] {
            k_fwdz_invz<N><<<dim3((unsigned)(n3_ / N / GS::LPW), (unsigned)nb), dim3(GS::THREADS), 0, stream_>>>(f);
        });
    });
    k_post_step<<<nb, 256, 0, stream_>>>(st_, c_, s0, partial_, ntiles_, tiles_z_, nt, ntau, stau);
    // streams that took the step: finish psi(t + dt) only as far as the next step needs it,
    // i.e. its density (and the ensemble density if asked for); psi itself is not written back
    a.active = st_.stepped;
    launch_psi<+1, 0, 0>(1, a, nb, PK_INV_Y);
    p.dens = step_dens_;
    p.active = st_.stepped;
    dispatch_n(N_, [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        using G0 = Geom<N, 0>;
        const dim3 grid((unsigned)(n3_ / N / G0::LPW)), block(G0::THREADS);
        timed(PK_INVX_RHO, [&] {
            if (step_dens_) k_invx_rho<N, 0, 0, 1><<<grid, block, 0, stream_>>>(p);
            else k_invx_rho<N, 0, 0, 0><<<grid, block, 0, stream_>>>(p);
        });
    });
    poisson_tail(s0, nb, false, true, true, st_.stepped, false, false);   // max|phi| is reduced by the next k_front
}

int Engine::step(std::string& err) {
    if (cfg_.expanding) { err = "expanding contexts step with msm_step_begin / msm_step_end"; return MSM_ERR_STATE; }
    if (in_step_) { err = "msm_step inside an open msm_step_begin"; return MSM_ERR_STATE; }
    phi_s0_ = -1;   // the batch potential buffer is not maintained by steps
    enter_fourier();
    if (int rc = mean_step_open(err)) return rc;
    auto body = [&]() -> hipError_t {
        if (step_dens_) { hipError_t e = hipMemsetAsync(step_dens_, 0, sizeof(double) * n3_, stream_); if (e != hipSuccess) return e; }
        for_batches([&](int s0, int nb) {
            step_front(s0, nb);
            step_back(s0, nb, nullptr, nullptr, nullptr);
        });
        publish();
        return hipSuccess;
    };
    // Small grids are bound by the host's launch rate (a step is ~25 launches of a few microseconds each): their step
    // is captured once into a HIP graph and replayed.  Every argument of every launch is constant from step to step
    // (dt, flags and counters live in device memory); only the set of batches that still have a live stream and the
    // density target can change, and either re-captures.
    if (graph_ok_ && !prof_on_ && !mean_on_ && n3_ * (size_t)ns_ <= GRAPH_MAX_CELLS) {
        std::vector<char> key;
        for (int s0 = 0; s0 < ns_; s0 += batch_) {
            char any = 0;
            for (int i = s0; i < std::min(ns_, s0 + batch_); ++i) any |= (char)(host_active_[i] != 0);
            key.push_back(any);
        }
        if (!graph_exec_ || key != graph_key_ || graph_dens_ != step_dens_) {
            if (graph_exec_) { (void)hipGraphExecDestroy(graph_exec_); graph_exec_ = nullptr; }
            hipGraph_t g = nullptr;
            hipError_t e = hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal);
            if (e == hipSuccess) {
                const hipError_t eb = body();
                e = hipStreamEndCapture(stream_, &g);
                if (e == hipSuccess) e = eb;
            }
            if (e == hipSuccess) e = hipGraphInstantiate(&graph_exec_, g, nullptr, nullptr, 0);
            if (g) (void)hipGraphDestroy(g);
            if (e != hipSuccess) {            // no graphs on this stream: plain launches from now on
                (void)hipGetLastError();
                graph_exec_ = nullptr;
                graph_ok_ = false;
            } else {
                graph_key_ = key;
                graph_dens_ = step_dens_;
            }
        }
        if (graph_exec_) {
            HIP_TRY(hipGraphLaunch(graph_exec_, stream_));
            return MSM_OK;
        }
    }
    HIP_TRY(body());
    HIP_TRY(hipGetLastError());
    return mean_step_close(err);
}

int Engine::step_begin(const double* a_now, const double* tau_next, std::string& err) {
    if (!cfg_.expanding) { err = "msm_step_begin is for expanding contexts"; return MSM_ERR_STATE; }
    if (in_step_) { err = "msm_step_begin called twice"; return MSM_ERR_STATE; }
    if (!a_now || !tau_next) { err = "null input"; return MSM_ERR_INVALID; }
    std::vector<real> a(ns_), t(ns_);
    for (int s = 0; s < ns_; ++s) { a[s] = (real)a_now[s]; t[s] = (real)tau_next[s]; }
    HIP_TRY(hipMemcpyAsync(st_.a_now, a.data(), sizeof(real) * ns_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(st_.tau_next, t.data(), sizeof(real) * ns_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));  // the host vectors die here
    enter_fourier();
    for_batches([&](int s0, int nb) { step_front(s0, nb); });
    publish();
    in_step_ = true;
    HIP_TRY(hipGetLastError());
    return MSM_OK;
}

int Engine::step_end(const double* kick_a, const double* new_time, const double* new_tau,
                     const double* snapped_tau, std::string& err) {
    if (!in_step_) { err = "msm_step_end without msm_step_begin"; return MSM_ERR_STATE; }
    if (!kick_a || !new_time || !new_tau || !snapped_tau) { err = "null input"; return MSM_ERR_INVALID; }
    HIP_TRY(hipStreamSynchronize(stream_));
    std::vector<real> k(ns_), nt(ns_), ntau(ns_), stau(ns_);
    for (int s = 0; s < ns_; ++s) {
        // two half-kicks exp(-i dtau/2 a1 phi) exp(-i dtau/2 a2 phi) on one phi (:726-742)
        const real dtau = (real)h_status_[s].dt;
        k[s] = -dtau / (real)2.0 * (real)kick_a[s];
        nt[s] = (real)new_time[s]; ntau[s] = (real)new_tau[s]; stau[s] = (real)snapped_tau[s];
    }
    HIP_TRY(hipMemcpyAsync(st_.kick, k.data(), sizeof(real) * ns_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(d_in_[0], nt.data(), sizeof(real) * ns_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(d_in_[1], ntau.data(), sizeof(real) * ns_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(d_in_[2], stau.data(), sizeof(real) * ns_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    phi_s0_ = -1;
    if (int rc = mean_step_open(err)) return rc;
    if (step_dens_) HIP_TRY(hipMemsetAsync(step_dens_, 0, sizeof(double) * n3_, stream_));
    for_batches([&](int s0, int nb) { step_back(s0, nb, d_in_[0], d_in_[1], d_in_[2]); });
    publish();
    in_step_ = false;
    HIP_TRY(hipGetLastError());
    return mean_step_close(err);
}

int Engine::sync(std::string& err) {
    HIP_TRY(hipStreamSynchronize(stream_));
    for (int s = 0; s < ns_; ++s) host_active_[s] = h_status_[s].active;
    return MSM_OK;
}

int Engine::get_status(msm_stream_status* out, int n, std::string& err) {
    if (!out || n < 0 || n > ns_) { err = "bad status buffer"; return MSM_ERR_INVALID; }
    HIP_TRY(hipStreamSynchronize(stream_));
    for (int s = 0; s < ns_; ++s) host_active_[s] = h_status_[s].active;
    std::memcpy(out, h_status_, sizeof(msm_stream_status) * n);
    return MSM_OK;
}

int Engine::potential(int first, int count, std::string& err) {
    if (first < 0 || count < 1 || first + count > ns_) { err = "stream range out of bounds"; return MSM_ERR_INVALID; }
    if (count > 2 * ((batch_ + 1) / 2)) { err = "count exceeds the batch size"; return MSM_ERR_INVALID; }
    if (in_step_) { err = "msm_potential inside an open msm_step_begin"; return MSM_ERR_STATE; }
    if (lowdim_) {
        if (fourier_) {
            fft3_plain(P_, false, X_, true, first, count, +1, nullptr);
            lowdim_poisson(X_, true, (real)1, first, count, true, all_active_);
        } else {
            lowdim_poisson(P_, false, (real)1, first, count, true, all_active_);
        }
    } else if (fourier_ && two_pass_) {
        {
            PoisArgs p = pois_args(first, count, all_active_);
            p.X = P_ + (size_t)first * n3_;
            p.scale = tp_scale_a_;
            timed(PK_INVX_RHO, [&] {
#ifdef MSM_TQ
                if (tq_) { tq_rho_from_state(p, count, false); return; }
#endif
                if (ts_) { ks_invx_rho<0><<<dim3(ts::A_TILES), dim3(ts::THREADS), 0, stream_>>>(p); return; }
                k_invx_rho3<0><<<dim3(tp::A_TILES), dim3(tp::A_THREADS), 0, stream_>>>(p);
            });
        }
        poisson_tail(first, count, true, true, true, all_active_);
    } else if (fourier_) {
        fft3_plain(P_, false, X_, true, first, count, +1, nullptr);
        poisson_from_psi(X_, true, first, count, true, true, all_active_);
    } else {
        poisson_from_psi(P_, false, first, count, true, true, all_active_);
    }
    publish(true);
    HIP_TRY(hipGetLastError());
    return MSM_OK;
}

int Engine::fft(int s, int dir, std::string& err) {
    if (int rc = check_stream(s, err)) return rc;
    if (int rc = need_real(err)) return rc;
    if (dir != MSM_FORWARD && dir != MSM_INVERSE) { err = "direction must be -1 or +1"; return MSM_ERR_INVALID; }
    fft3_plain(P_, false, P_, false, s, 1, dir, nullptr);
    HIP_TRY(hipGetLastError());
    return MSM_OK;
}

int Engine::set_psi(int s, const void* re, const void* im, std::string& err) {
    if (int rc = check_stream(s, err)) return rc;
    if (int rc = need_real(err)) return rc;
    if (!re || !im) { err = "null plane"; return MSM_ERR_INVALID; }
    real* stage = (real*)X_;   // two real planes fit one complex scratch grid
    HIP_TRY(hipMemcpyAsync(stage, re, sizeof(real) * n3_, hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(stage + n3_, im, sizeof(real) * n3_, hipMemcpyHostToDevice, stream_));
    k_interleave<<<2048, 256, 0, stream_>>>(P_ + (size_t)s * n3_, stage, stage + n3_, n3_);
    HIP_TRY(hipStreamSynchronize(stream_));
    return MSM_OK;
}

int Engine::copy_psi(int dst, int src, std::string& err) {
    if (int rc = check_stream(dst, err)) return rc;
    if (int rc = check_stream(src, err)) return rc;
    if (int rc = need_real(err)) return rc;
    if (dst == src) return MSM_OK;
    HIP_TRY(hipMemcpyAsync(P_ + (size_t)dst * n3_, P_ + (size_t)src * n3_, sizeof(cplx) * n3_, hipMemcpyDeviceToDevice, stream_));
    return MSM_OK;
}

int Engine::sample(int s, int scheme, uint64_t seed, std::string& err) {
    if (int rc = check_stream(s, err)) return rc;
    if (int rc = need_real(err)) return rc;
    if (scheme != MSM_WIGNER && scheme != MSM_HUSIMI && scheme != MSM_POISSON) { err = "unknown sampling scheme"; return MSM_ERR_INVALID; }
    if (scheme == MSM_POISSON) {
        k_sample_poisson<<<2048, 256, 0, stream_>>>(P_ + (size_t)s * n3_, n3_, (unsigned long long)seed,
                                                    std::pow((double)dx_, (double)dims_), cfg_.total_mass / cfg_.particle_mass);
        HIP_TRY(hipGetLastError());
        return MSM_OK;
    }
    const double n = cfg_.total_mass / cfg_.particle_mass;      // ics.rs:472
    const real sqrt_n = (real)std::sqrt(n);
    const real sqrt_dv = std::sqrt(std::pow(dx_, (real)dims_));  // ics.rs:479-489
    const real div = scheme == MSM_WIGNER ? sqrt_n * (real)2.0 : sqrt_n * std::sqrt((real)2.0);
    k_sample<<<2048, 256, 0, stream_>>>(P_ + (size_t)s * n3_, n3_, (unsigned long long)seed, sqrt_dv, div);
    HIP_TRY(hipGetLastError());
    return MSM_OK;
}

// psi (or psi_k in the reference's element order) of stream s as two real planes in `stage`, on the compute stream
int Engine::planes_of(int s, bool kspace, real* stage, std::string& err) {
    if (int rc = check_stream(s, err)) return rc;
    if (in_step_) { err = "download inside an open msm_step_begin"; return MSM_ERR_STATE; }
    if (kspace && !fourier_) { err = "psi_k is only defined after the first step"; return MSM_ERR_STATE; }
    const cplx* src = P_ + (size_t)s * n3_;
    if (!kspace && fourier_) {
        // psi = F^-1 psi_k (simulation_object.rs:581), materialised on demand in the batch scratch
        if (two_pass_) {
            psi_a(+1, src, true, W_, true, s, 1, nullptr);
            src = W_;
        } else {
            fft3_plain(P_, false, X_, true, s, 1, +1, nullptr);
            src = X_;
        }
    } else if (kspace && two_pass_) {
        const unsigned nb32 = (unsigned)(N_ / 32);
        psi_b_fwd(src, X_, s, 1);
        k_transpose_xz<<<dim3(nb32, nb32, (unsigned)N_), 256, 0, stream_>>>(X_, W_, N_);
        src = W_;
    }
    k_split<<<2048, 256, 0, stream_>>>(src, stage, stage + n3_, n3_);
    HIP_TRY(hipGetLastError());
    return MSM_OK;
}

int Engine::get_psi(int s, void* re, void* im, bool kspace, std::string& err) {
    if (!re || !im) { err = "null plane"; return MSM_ERR_INVALID; }
    real* stage = (real*)stage_;   // two real planes fit one complex grid
    if (int rc = planes_of(s, kspace, stage, err)) return rc;
    HIP_TRY(hipMemcpyAsync(re, stage, sizeof(real) * n3_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipMemcpyAsync(im, stage + n3_, sizeof(real) * n3_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return MSM_OK;
}

// Asynchronous download: the planes are formed on the compute stream into one of DL_SLOTS staging grids and
// copied out on a second stream, so that the steps that follow overlap the PCIe transfer (pinned host
// buffers) and the caller's file writers.
int Engine::get_psi_async(int s, void* re, void* im, int* ticket, std::string& err) {
    if (!re || !im || !ticket) { err = "null plane or ticket"; return MSM_ERR_INVALID; }
    if (!copy_stream_) {
        HIP_TRY(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
        for (int i = 0; i < DL_SLOTS; ++i) {
            HIP_TRY(hipMalloc((void**)&dl_stage_[i], sizeof(cplx) * n3_));
            HIP_TRY(hipEventCreateWithFlags(&dl_ready_[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&dl_done_[i], hipEventDisableTiming));
            HIP_TRY(hipEventRecord(dl_done_[i], copy_stream_));
        }
    }
    const int slot = dl_next_ % DL_SLOTS;
    HIP_TRY(hipEventSynchronize(dl_done_[slot]));   // the staging grid's previous copy has left
    real* stage = (real*)dl_stage_[slot];
    if (int rc = planes_of(s, false, stage, err)) return rc;
    dl_next_ += 1;
    HIP_TRY(hipEventRecord(dl_ready_[slot], stream_));
    HIP_TRY(hipStreamWaitEvent(copy_stream_, dl_ready_[slot], 0));
    HIP_TRY(hipMemcpyAsync(re, stage, sizeof(real) * n3_, hipMemcpyDeviceToHost, copy_stream_));
    HIP_TRY(hipMemcpyAsync(im, stage + n3_, sizeof(real) * n3_, hipMemcpyDeviceToHost, copy_stream_));
    HIP_TRY(hipEventRecord(dl_done_[slot], copy_stream_));
    *ticket = slot;
    return MSM_OK;
}

int Engine::download_wait(int ticket, std::string& err) {
    if (ticket < 0 || ticket >= DL_SLOTS || !copy_stream_) { err = "bad download ticket"; return MSM_ERR_INVALID; }
    HIP_TRY(hipEventSynchronize(dl_done_[ticket]));
    return MSM_OK;
}

// Restart support (the reference has none; its `time` key assumes t0 = 0, simulation_object.rs:627,915): before
// the first step, place a stream at (time, tau) with `current_dumps` dumps already written.
int Engine::set_clock(int s, double time, double tau, int current_dumps, std::string& err) {
    if (int rc = check_stream(s, err)) return rc;
    if (int rc = need_real(err)) return rc;
    if (current_dumps < 0 || current_dumps > (int)cfg_.num_data_dumps) { err = "current_dumps out of range"; return MSM_ERR_INVALID; }
    const real t = (real)time, ta = (real)tau;
    const int act = t < c_.tfinal ? 1 : 0;
    HIP_TRY(hipMemcpyAsync(st_.time + s, &t, sizeof(real), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(st_.tau + s, &ta, sizeof(real), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(st_.dumps + s, &current_dumps, sizeof(int), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipMemcpyAsync(st_.active + s, &act, sizeof(int), hipMemcpyHostToDevice, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    host_active_[s] = act;
    publish();
    HIP_TRY(hipStreamSynchronize(stream_));
    return MSM_OK;
}

int Engine::get_potential(int s, void* phi, std::string& err) {
    if (int rc = check_stream(s, err)) return rc;
    if (!phi) { err = "null plane"; return MSM_ERR_INVALID; }
    if (phi_s0_ < 0 || s < phi_s0_ || s >= phi_s0_ + phi_nb_) {
        err = "potential of this stream is not resident (call msm_potential for its batch)";
        return MSM_ERR_STATE;
    }
    HIP_TRY(hipMemcpyAsync(phi, PHI_ + (size_t)(s - phi_s0_) * n3_, sizeof(real) * n3_, hipMemcpyDeviceToHost, stream_));
    HIP_TRY(hipStreamSynchronize(stream_));
    return MSM_OK;
}

int Engine::density_sum(void* dev_out, std::string& err) {
    if (!dev_out) { err = "null output"; return MSM_ERR_INVALID; }
    if (in_step_) { err = "msm_density_sum inside an open msm_step_begin"; return MSM_ERR_STATE; }
    if (!fourier_) {
        k_density_sum<<<2048, 256, 0, stream_>>>(P_, n3_, ns_, (double*)dev_out, 0);
    } else if (two_pass_) {
        for (int s = 0; s < ns_; ++s) {
            psi_a(+1, P_ + (size_t)s * n3_, true, W_, true, s, 1, nullptr);
            k_density_sum<<<2048, 256, 0, stream_>>>(W_, n3_, 1, (double*)dev_out, s > 0 ? 1 : 0);
        }
    } else {
        for (int s0 = 0; s0 < ns_; s0 += batch_) {
            const int nb = std::min(batch_, ns_ - s0);
            fft3_plain(P_, false, X_, true, s0, nb, +1, nullptr);
            k_density_sum<<<2048, 256, 0, stream_>>>(X_, n3_, nb, (double*)dev_out, s0 > 0 ? 1 : 0);
        }
    }
    HIP_TRY(hipGetLastError());
    return MSM_OK;
}

int Engine::step_density(void* dev_out, std::string& err) {
    if (mean_on_) { err = "the step's density target belongs to msm_mean_density_begin on this context"; return MSM_ERR_STATE; }
    step_dens_ = (double*)dev_out;
    return MSM_OK;
}

// ---------------------------------------------------------------------------
// several GPUs
// ---------------------------------------------------------------------------
int Engine::comm_init(const void* id, int rank, int world, std::string& err) {
    if (comm_) { err = "msm_comm_init called twice"; return MSM_ERR_STATE; }
    if (!id || world < 1 || rank < 0 || rank >= world) { err = "bad communicator arguments"; return MSM_ERR_INVALID; }
    rccl_ = rccl(err);
    if (!rccl_) return MSM_ERR_INVALID;
    CommId cid;
    std::memcpy(cid.bytes, id, COMM_ID_BYTES);
    const int rc = rccl_->comm_init_rank(&comm_, world, cid, rank);   // meets the other ranks
    if (rc != 0) { comm_ = nullptr; return nccl_fail(rc, "ncclCommInitRank", err); }
    comm_rank_ = rank; comm_world_ = world;
    // all or nothing: a failure below leaves the context without a communicator, so that a second call is a clean
    // retry instead of "called twice" on half-built resources
    const int rs = [&]() -> int {
        HIP_TRY(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&comm_ev_, hipEventDisableTiming));
        HIP_TRY(hipMalloc((void**)&comm_small_d_, sizeof(double) * COMM_SMALL));
        HIP_TRY(hipHostMalloc((void**)&comm_small_h_, sizeof(double) * COMM_SMALL));
        return MSM_OK;
    }();
    if (rs != MSM_OK) {
        (void)rccl_->comm_destroy(comm_);
        comm_ = nullptr;
        if (comm_small_d_) { (void)hipFree(comm_small_d_); comm_small_d_ = nullptr; }
        if (comm_small_h_) { (void)hipHostFree(comm_small_h_); comm_small_h_ = nullptr; }
        if (comm_ev_) { (void)hipEventDestroy(comm_ev_); comm_ev_ = nullptr; }
        if (comm_stream_) { (void)hipStreamDestroy(comm_stream_); comm_stream_ = nullptr; }
    }
    return rs;
}

int Engine::mean_density_begin(long long total_streams, std::string& err) {
    if (!comm_) { err = "msm_comm_init first"; return MSM_ERR_STATE; }
    if (in_step_) { err = "inside an open msm_step_begin"; return MSM_ERR_STATE; }
    if (total_streams < 1) { err = "total_streams must be >= 1"; return MSM_ERR_INVALID; }
    mean_total_ = total_streams;
    if (mean_on_) return MSM_OK;
    // resources that a failed earlier call has already created are kept and reused, never allocated twice
    for (int i = 0; i < 2; ++i) {
        if (!mean_buf_[i]) HIP_TRY(hipMalloc((void**)&mean_buf_[i], sizeof(double) * n3_));
        if (!dens_ready_[i]) HIP_TRY(hipEventCreateWithFlags(&dens_ready_[i], hipEventDisableTiming));
        if (!comm_done_[i]) HIP_TRY(hipEventCreateWithFlags(&comm_done_[i], hipEventDisableTiming));
    }
    if (!mean_out_) HIP_TRY(hipMalloc((void**)&mean_out_, sizeof(double) * n3_));
    mean_on_ = true;
    mean_next_ = 0; mean_last_ = -1;
    step_dens_ = nullptr;
    return MSM_OK;
}

int Engine::mean_step_open(std::string& err) {
    if (!mean_on_) return MSM_OK;
    // this buffer's previous all-reduce (two steps ago) must have left it before the step clears it
    HIP_TRY(hipStreamWaitEvent(stream_, comm_done_[mean_next_], 0));
    step_dens_ = mean_buf_[mean_next_];
    return MSM_OK;
}

int Engine::mean_step_close(std::string& err) {
    if (!mean_on_) return MSM_OK;
    const int b = mean_next_;
    HIP_TRY(hipEventRecord(dens_ready_[b], stream_));
    HIP_TRY(hipStreamWaitEvent(comm_stream_, dens_ready_[b], 0));
    // sum over the GPUs of sum_s |psi_s|^2 (synthesizer/src/lib.rs:217-229 in situ); runs beside the next step
    const int rc = rccl_->all_reduce(mean_buf_[b], mean_buf_[b], n3_, COMM_F64, COMM_SUM, comm_, comm_stream_);
    if (rc != 0) return nccl_fail(rc, "ncclAllReduce", err);
    HIP_TRY(hipEventRecord(comm_done_[b], comm_stream_));
    mean_last_ = b;
    mean_next_ = b ^ 1;
    return MSM_OK;
}

int Engine::mean_density(double* host, void** dev, std::string& err) {
    if (!mean_on_) { err = "msm_mean_density_begin first"; return MSM_ERR_STATE; }
    if (mean_last_ < 0) { err = "no step has been taken since msm_mean_density_begin"; return MSM_ERR_STATE; }
    k_scale_f64<<<2048, 256, 0, comm_stream_>>>(mean_buf_[mean_last_], mean_out_, n3_, 1.0 / (double)mean_total_);
    if (host) HIP_TRY(hipMemcpyAsync(host, mean_out_, sizeof(double) * n3_, hipMemcpyDeviceToHost, comm_stream_));
    HIP_TRY(hipStreamSynchronize(comm_stream_));
    if (dev) *dev = mean_out_;
    return MSM_OK;
}

// Every collective of a context is enqueued on its communication stream, in the order the host issues them: one
// communicator, one stream.  The compute stream meets it through events only.
int Engine::comm_allreduce_f64(double* vals, int n, int op, std::string& err) {
    if (!comm_) { err = "msm_comm_init first"; return MSM_ERR_STATE; }
    if (!vals || n < 1 || n > COMM_SMALL) { err = "1 to 64 values"; return MSM_ERR_INVALID; }
    if (op != COMM_SUM && op != COMM_MAX && op != COMM_MIN) { err = "op must be sum (0), max (2) or min (3)"; return MSM_ERR_INVALID; }
    std::memcpy(comm_small_h_, vals, sizeof(double) * n);
    HIP_TRY(hipMemcpyAsync(comm_small_d_, comm_small_h_, sizeof(double) * n, hipMemcpyHostToDevice, comm_stream_));
    const int rc = rccl_->all_reduce(comm_small_d_, comm_small_d_, (size_t)n, COMM_F64, op, comm_, comm_stream_);
    if (rc != 0) return nccl_fail(rc, "ncclAllReduce", err);
    HIP_TRY(hipMemcpyAsync(comm_small_h_, comm_small_d_, sizeof(double) * n, hipMemcpyDeviceToHost, comm_stream_));
    HIP_TRY(hipStreamSynchronize(comm_stream_));
    std::memcpy(vals, comm_small_h_, sizeof(double) * n);
    return MSM_OK;
}

// The synthesizer's sums of one dump index over all GPUs (the reference gathers per-rank results over MPI,
// synthesizer/src/lib.rs:466-583): one all-reduce of the four fields, one of the stream count.
int Engine::synth_allreduce(int slot, std::string& err) {
    if (!comm_) { err = "msm_comm_init first"; return MSM_ERR_STATE; }
    if (slot < 0 || slot > 65535) { err = "synthesizer slot out of range"; return MSM_ERR_INVALID; }
    if ((int)synth_.size() <= slot) synth_.resize(slot + 1);
    SynthSlot& sl = synth_[slot];
    if (!sl.sums) {   // a rank without seed streams contributes zeros
        HIP_TRY(hipMalloc((void**)&sl.sums, sizeof(double2) * 4 * n3_));
        HIP_TRY(hipMemsetAsync(sl.sums, 0, sizeof(double2) * 4 * n3_, stream_));
        sl.count = 0;
    }
    // the accumulations were enqueued on the compute stream: the collective starts behind them
    HIP_TRY(hipEventRecord(comm_ev_, stream_));
    HIP_TRY(hipStreamWaitEvent(comm_stream_, comm_ev_, 0));
    int rc = rccl_->all_reduce(sl.sums, sl.sums, 8 * n3_, COMM_F64, COMM_SUM, comm_, comm_stream_);
    if (rc != 0) return nccl_fail(rc, "ncclAllReduce", err);
    double c = (double)sl.count;
    if (int r2 = comm_allreduce_f64(&c, 1, COMM_SUM, err)) return r2;   // waits for the communication stream: the sums are final
    sl.count = (long long)(c + 0.5);
    return MSM_OK;
}

int Engine::profile(int enable, std::string& err) {
    HIP_TRY(hipStreamSynchronize(stream_));
    for (auto& r : prof_) { ev_pool_.push_back(r.a); ev_pool_.push_back(r.b); }
    prof_.clear();
    prof_on_ = enable != 0;
    return MSM_OK;
}

int Engine::profile_report(int max_kinds, double* total_ms, long long* launches, int* n_kinds, std::string& err) {
    if (!total_ms || !launches || !n_kinds || max_kinds < PK_COUNT) { err = "report buffers too small"; return MSM_ERR_INVALID; }
    HIP_TRY(hipStreamSynchronize(stream_));
    for (int k = 0; k < PK_COUNT; ++k) { total_ms[k] = 0.0; launches[k] = 0; }
    for (auto& r : prof_) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, r.a, r.b));
        total_ms[r.kind] += ms;
        launches[r.kind] += 1;
        ev_pool_.push_back(r.a);
        ev_pool_.push_back(r.b);
    }
    prof_.clear();
    *n_kinds = PK_COUNT;
    return MSM_OK;
}

void Engine::materialise(int s, const cplx** psi, const cplx** psik) {
    if (!fourier_) {
        // before the first step P_ holds psi; psi_k = F psi (simulation_object.rs:477-479) into the scratch
        fft3_plain(P_, false, X_, true, s, 1, -1, nullptr);
        *psi = P_ + (size_t)s * n3_;
        *psik = X_;
    } else if (two_pass_) {
        const cplx* S = P_ + (size_t)s * n3_;
        psi_a(+1, S, true, W_, true, s, 1, nullptr);
        psi_b_fwd(S, X_, s, 1);
        const unsigned nb32 = (unsigned)(N_ / 32);
        k_transpose_xz<<<dim3(nb32, nb32, (unsigned)N_), 256, 0, stream_>>>(X_, stage_, N_);
        *psi = W_;
        *psik = stage_;
    } else {
        fft3_plain(P_, false, X_, true, s, 1, +1, nullptr);
        *psi = X_;
        *psik = P_ + (size_t)s * n3_;
    }
}

int Engine::synth_accumulate(int s, int slot, std::string& err) {
    if (int rc = check_stream(s, err)) return rc;
    if (slot < 0 || slot > 65535) { err = "synthesizer slot out of range"; return MSM_ERR_INVALID; }
    if (in_step_) { err = "msm_synth_accumulate inside an open msm_step_begin"; return MSM_ERR_STATE; }
    if ((int)synth_.size() <= slot) synth_.resize(slot + 1);
    SynthSlot& sl = synth_[slot];
    if (!sl.sums) {
        HIP_TRY(hipMalloc((void**)&sl.sums, sizeof(double2) * 4 * n3_));
        HIP_TRY(hipMemsetAsync(sl.sums, 0, sizeof(double2) * 4 * n3_, stream_));
        sl.count = 0;
    }
    const cplx *psi = nullptr, *psik = nullptr;
    materialise(s, &psi, &psik);
    k_synth_acc<<<2048, 256, 0, stream_>>>(psi, psik, sl.sums, n3_, std::pow((double)N_, dims_ / 2.0));
    sl.count += 1;
    HIP_TRY(hipGetLastError());
    return MSM_OK;
}

int Engine::synth_device(int slot, void** dev, long long* count, std::string& err) {
    if (int rc = synth_check(slot, err)) return rc;
    HIP_TRY(hipStreamSynchronize(stream_));   // the caller hands the buffer to another library
    if (dev) *dev = synth_[slot].sums;
    if (count) *count = synth_[slot].count;
    return MSM_OK;
}

int Engine::synth_set_count(int slot, long long count, std::string& err) {
    if (int rc = synth_check(slot, err)) return rc;
    if (count < 1) { err = "count must be >= 1"; return MSM_ERR_INVALID; }
    synth_[slot].count = count;
    return MSM_OK;
}

int Engine::synth_fetch(int slot, int field, double* re, double* im, std::string& err) {
    if (int rc = synth_check(slot, err)) return rc;
    if (field < 0 || field > 3 || !re || !im) { err = "bad field or null plane"; return MSM_ERR_INVALID; }
    if (synth_[slot].count < 1) { err = "nothing accumulated"; return MSM_ERR_STATE; }
    double* stage = nullptr;   // two double planes: fp32 contexts have a smaller staging grid, so use a scratch of our own
    HIP_TRY(hipMalloc((void**)&stage, sizeof(double) * 2 * n3_));
    k_synth_mean<<<2048, 256, 0, stream_>>>(synth_[slot].sums + (size_t)field * n3_, stage, stage + n3_, n3_,
                                            1.0 / (double)synth_[slot].count);
    hipError_t e1 = hipMemcpyAsync(re, stage, sizeof(double) * n3_, hipMemcpyDeviceToHost, stream_);
    hipError_t e2 = hipMemcpyAsync(im, stage + n3_, sizeof(double) * n3_, hipMemcpyDeviceToHost, stream_);
    hipError_t e3 = hipStreamSynchronize(stream_);
    (void)hipFree(stage);
    HIP_TRY(e1); HIP_TRY(e2); HIP_TRY(e3);
    return MSM_OK;
}

int Engine::synth_qx(int slot, double* re, double* im, std::string& err) {
    if (int rc = synth_check(slot, err)) return rc;
    if (!re || !im) { err = "null output"; return MSM_ERR_INVALID; }
    if (synth_[slot].count < 1) { err = "nothing accumulated"; return MSM_ERR_STATE; }
    constexpr int NB = 1024;
    double2* part = nullptr;
    HIP_TRY(hipMalloc((void**)&part, sizeof(double2) * NB));
    k_synth_qx<<<NB, 256, 0, stream_>>>(synth_[slot].sums, n3_, 1.0 / (double)synth_[slot].count, part);
    std::vector<double2> h(NB);
    hipError_t e1 = hipMemcpyAsync(h.data(), part, sizeof(double2) * NB, hipMemcpyDeviceToHost, stream_);
    hipError_t e2 = hipStreamSynchronize(stream_);
    (void)hipFree(part);
    HIP_TRY(e1); HIP_TRY(e2);
    double sr = 0.0, si = 0.0;
    for (int i = 0; i < NB; ++i) { sr += h[i].x; si += h[i].y; }   // fixed order
    const double dv = std::pow((double)dx_, (double)dims_);
    *re = sr * dv;
    *im = si * dv;
    return MSM_OK;
}

// The same product without any allocation, synchronisation or host copy: partials go to this device's fixed
// workspace, a one-thread kernel adds them in the order synth_qx's host loop does, the result lands in the
// caller's device-accessible buffer when the context's stream gets there.
int Engine::synth_qx_noalloc(int slot, double* out2, std::string& err) {
    if (int rc = synth_check(slot, err)) return rc;
    if (!out2) { err = "null output"; return MSM_ERR_INVALID; }
    if (synth_[slot].count < 1) { err = "nothing accumulated"; return MSM_ERR_STATE; }
    if (n3_ > QX_MAX_CELLS) { err = "grid exceeds the fixed workspace's limit of 2^30 cells"; return MSM_ERR_INVALID; }
    k_synth_qx<<<QX_PARTS, 256, 0, stream_>>>(synth_[slot].sums, n3_, 1.0 / (double)synth_[slot].count, qx_scratch_);
    k_qx_finish<<<1, 64, 0, stream_>>>(qx_scratch_, QX_PARTS, std::pow((double)dx_, (double)dims_), out2);
    HIP_TRY(hipGetLastError());
    return MSM_OK;
}

int Engine::synth_release(int slot, std::string& err) {
    if (int rc = synth_check(slot, err)) return rc;
    HIP_TRY(hipStreamSynchronize(stream_));
    (void)hipFree(synth_[slot].sums);
    synth_[slot] = SynthSlot{};
    return MSM_OK;
}

int Engine::timer_start(std::string& err) {
    HIP_TRY(hipEventRecord(ev0_, stream_));
    return MSM_OK;
}
int Engine::timer_stop(float* ms, std::string& err) {
    if (!ms) { err = "null output"; return MSM_ERR_INVALID; }
    HIP_TRY(hipEventRecord(ev1_, stream_));
    HIP_TRY(hipEventSynchronize(ev1_));
    HIP_TRY(hipEventElapsedTime(ms, ev0_, ev1_));
    return MSM_OK;
}

}  // namespace

#define MSM_CAT2(a, b) a##b
#define MSM_CAT(a, b) MSM_CAT2(a, b)
IEngine* MSM_CAT(make_engine_, MSM_SUFFIX)(const msm_config& cfg) { return new Engine(cfg); }

}  // namespace msm
