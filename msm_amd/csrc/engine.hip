// Split-step Schroedinger-Poisson engine for gfx950, one real type per
// translation unit (-DMSM_REAL=double|float -DMSM_SUFFIX=f64|f32).
//
// Device layout (per context):
//   P[n_streams][N^3]  complex  the state: psi before the first step, psi_k
//                               afterwards                 simulation_object.rs:57,60
//   X[batch][N^3]      complex  psi of the current batch inside a step (scratch)
//   W[batch/2][N^3]    complex  Poisson work array: two streams' densities
//                               packed as re/im of one complex grid
//   PHI[batch][N^3]    real     potential per stream     simulation_object.rs:63
// A 3-D transform is three passes of the batched line FFT (fft_core.hpp), one
// per axis, each moving every element once HBM -> registers -> HBM; all
// pointwise operators of the step are applied inside a pass's load or store,
// and neighbouring passes that work on the same lines are fused (forward z +
// inverse z around the second drift; inverse x + density + Poisson forward x).
// Between steps only psi_k is kept; psi is rebuilt on demand for downloads.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

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

template <int THREADS>
__device__ __forceinline__ double block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int off = THREADS / 2; off > 0; off >>= 1) {
        if (tid < off) red[tid] += red[tid + off];
        __syncthreads();
    }
    return red[0];
}
template <int THREADS>
__device__ __forceinline__ double block_max(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
#pragma unroll
    for (int off = THREADS / 2; off > 0; off >>= 1) {
        // NaN must win so that a poisoned potential is visible to the host
        if (tid < off) {
            double a = red[tid], b = red[tid + off];
            red[tid] = (a != a) ? a : ((b != b) ? b : (a > b ? a : b));
        }
        __syncthreads();
    }
    return red[0];
}

// 1/x for the Green's function: hardware reciprocal plus one Newton step (about 1 ulp in double),
// a third of the instructions of the IEEE division expansion.
__device__ __forceinline__ double recip_t(double x) {
    const double r = __builtin_amdgcn_rcp(x);
    return fma(fma(-x, r, 1.0), r, r);
}
__device__ __forceinline__ float recip_t(float x) { return 1.0f / x; }

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
__device__ __forceinline__ double nanmax(double a, double b) { return (a != a) ? a : ((b != b) ? b : (a > b ? a : b)); }
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
        __syncthreads();
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

// Waves per SIMD at which a generic pass kernel fills the CU as far as its LDS tile allows: the register allocator is
// told to stay within 512 / that many VGPRs, so that the workgroups the LDS admits really are resident together
// (in single precision the tiles are half as large and the kernels otherwise settle on one workgroup per CU).
template <int N, int AX>
constexpr int occ_waves() {
    using G = Geom<N, AX>;
    const int bytes = (int)(G::LDS_ELEMS * sizeof(cplx)) + 4096;      // tile + twiddles + reduction scratch
    const int wpw = G::THREADS >= 64 ? G::THREADS / 64 : 1;          // waves per workgroup
    int wgs = 163840 / bytes;
    if (wgs > 32 / wpw) wgs = 32 / wpw;
    if (wgs < 1) wgs = 1;
    int w = wgs * wpw / 4;
    return w < 1 ? 1 : (w > 4 ? 4 : w);
}
#define MSM_OCC(N, AX) __attribute__((amdgpu_waves_per_eu(occ_waves<N, AX>())))

// ---------------------------------------------------------------------------
// generic axis pass: transforms of psi / psi_k and the plain passes on W
// ---------------------------------------------------------------------------
struct PsiArgs {
    const cplx* src;
    cplx* dst;
    size_t n3;
    const int* active;   // may be null
    int s0;              // first (global) stream of this launch
    int src_local;       // 1: src is indexed by blockIdx.y (batch scratch), 0: by the global stream id
    int dst_local;
    int pair_nb;         // > 0: blockIdx.y is a PAIR index over nb streams (W passes)
    int nlines;          // x passes: number of lines of the grid (a 1-D grid has fewer than one tile); 0 = no limit
    const cplx* tw;
    const cplx* drift;   // [n_streams][N]
    const real* phi;     // [batch][n3], local stream index
    const real* kick;    // [n_streams]
    real scale;
};

// LOAD: 0 plain, 1 multiply by the drift table at the line position (simulation_object.rs:504-516),
//       2 kick psi *= exp(i kick phi)                                 (simulation_object.rs:535-545)
// STORE: 0 scale only, 1 drift table times scale                      (simulation_object.rs:562-574)
template <int N, int SGN, int AX, int LOAD, int STORE>
__global__ __launch_bounds__((Geom<N, AX>::THREADS)) MSM_OCC(N, AX) void k_psi_pass(PsiArgs a) {
    using G = Geom<N, AX>;
    constexpr int E = G::E, TPL = G::TPL;
    __shared__ cplx lds[G::LDS_ELEMS];

    const int ls = blockIdx.y;
    const int sg = a.s0 + ls;  // global stream id (per-stream tables, activity)
    int si = a.src_local ? ls : sg, di = a.dst_local ? ls : sg;
    if (a.pair_nb > 0) {
        const int sa = a.s0 + 2 * ls, sb = sa + 1;
        const bool act = a.active == nullptr || a.active[sa] || (2 * ls + 1 < a.pair_nb && a.active[sb]);
        if (!act) return;
        si = di = ls;  // W is indexed by the local pair
    } else if (a.active != nullptr && !a.active[sg]) {
        return;
    }

    const Tile<N, AX> t(blockIdx.x, threadIdx.x);
    cplx* tw = lds + G::LPW * G::LINE_LDS;
    load_twiddles<N, AX>(tw, a.tw);
    cplx* line = lds + t.ln * G::LINE_LDS;

    const cplx* __restrict__ src = a.src + (size_t)si * a.n3;
    cplx* __restrict__ dst = a.dst + (size_t)di * a.n3;

    // a tile past the end of a small grid (x passes of 1-D grids): its threads transform zeros and store nothing
    const bool valid = AX != 0 || a.nlines == 0 || (int)(blockIdx.x * G::LPW + t.ln) < a.nlines;
    cplx v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) v[q] = valid ? src[t.base + (size_t)(t.j0 + TPL * q) * t.stride] : mk<real>(0, 0);

    if constexpr (LOAD == 1) {
        const cplx* __restrict__ d = a.drift + (size_t)sg * N;
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = cmul(v[q], d[t.j0 + TPL * q]);
    } else if constexpr (LOAD == 2) {
        const real* __restrict__ phi = a.phi + (size_t)ls * a.n3;
        const real kc = a.kick[sg];
        real ph[E];
#pragma unroll
        for (int q = 0; q < E; ++q) ph[q] = valid ? phi[t.base + (size_t)(t.j0 + TPL * q) * t.stride] : (real)0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            real sn, cs;
            sincos_t(kc * ph[q], &sn, &cs);
            v[q] = cmul(v[q], mk<real>(cs, sn));
        }
    }

    LineFft<real, N, SGN, AX == 0>::run(v, t.j0, line, tw);

    if (valid) {
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const int pos = t.j0 + TPL * q;
            cplx val = v[q];
            if constexpr (STORE >= 1) val = cmul(val, a.drift[(size_t)sg * N + pos]);
            dst[t.base + (size_t)pos * t.stride] = cscale(val, a.scale);
        }
    }
}

// ---------------------------------------------------------------------------
// fused z pass around the second kinetic half-step:
//   psi_k = D_z F_z[.] / N^{3/2}   -> stored (the state carried to the next step), aliased-mass sum,
//   then straight on with F_z^{-1} of the same lines (first pass of psi = F^-1 psi_k).
// (simulation_object.rs:552-581 and check_alias :1249-1280.)  Saves one read of psi_k.
// ---------------------------------------------------------------------------
struct FzArgs {
    const cplx* X;       // [batch][n3] local: input (x and y already transformed) and output
    cplx* Xout;
    cplx* P;             // [n_streams][n3] global: psi_k
    size_t n3;
    const int* active;
    int s0;
    const cplx* tw;
    const cplx* drift;
    real scale;          // 1 / N^{dims/2}
    const real* ksq;
    real alias_cut, fourpi2;
    double* partial;     // [n_streams][ntiles]
    int ntiles;
};

template <int N>
__global__ __launch_bounds__((Geom<N, 2>::THREADS)) MSM_OCC(N, 2) void k_fwdz_invz(FzArgs a) {
    using G = Geom<N, 2>;
    constexpr int E = G::E, TPL = G::TPL;
    __shared__ cplx lds[G::LDS_ELEMS];
    __shared__ double red[G::THREADS];

    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (!a.active[sg]) return;
    const Tile<N, 2> t(blockIdx.x, threadIdx.x);
    cplx* tw = lds + G::LPW * G::LINE_LDS;
    load_twiddles<N, 2>(tw, a.tw);
    cplx* line = lds + t.ln * G::LINE_LDS;
    const cplx* __restrict__ x = a.X + (size_t)ls * a.n3;
    cplx* __restrict__ xo = a.Xout + (size_t)ls * a.n3;
    cplx* __restrict__ pk = a.P + (size_t)sg * a.n3;
    const cplx* __restrict__ d = a.drift + (size_t)sg * N;

    cplx v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) v[q] = x[t.base + (size_t)(t.j0 + TPL * q) * t.stride];
    LineFft<real, N, -1>::run(v, t.j0, line, tw);

    const real kxy = a.ksq[t.c1] + a.ksq[t.c2];  // ((kx^2 + ky^2) + kz^2) * (2 pi)^2, utils/fft.rs:140-160
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const int pos = t.j0 + TPL * q;
        v[q] = cscale(cmul(v[q], d[pos]), a.scale);
        pk[t.base + (size_t)pos * t.stride] = v[q];
        const real k2 = (kxy + a.ksq[pos]) * a.fourpi2;
        if (k2 > a.alias_cut) acc += (double)(v[q].x * v[q].x + v[q].y * v[q].y);
    }
    const double tot = block_sum<G::THREADS>(acc, red);   // also orders the LDS reuse below
    if (threadIdx.x == 0) a.partial[(size_t)sg * a.ntiles + blockIdx.x] = tot;

    LineFft<real, N, +1>::run(v, t.j0, line, tw);
#pragma unroll
    for (int q = 0; q < E; ++q) xo[t.base + (size_t)(t.j0 + TPL * q) * t.stride] = v[q];
}

// ---------------------------------------------------------------------------
// Poisson solve (calculate_potential, simulation_object.rs:1066-1110) on
// stream pairs: rho_a + i rho_b is transformed as one complex grid; the
// Green's function is real and even, so the inverse returns phi_a + i phi_b.
// ---------------------------------------------------------------------------
struct PoisArgs {
    const cplx* X;       // psi, [.][n3]
    cplx* Xout;          // where the fused inverse pass stores psi (may alias X)
    int x_local;         // 1: X is the batch scratch (local index), 0: the per-stream state
    cplx* W;             // [pairs][n3]
    real* PHI;           // [batch][n3]
    size_t n3;
    const int* active;
    int s0, nb;
    const cplx* tw;
    const cplx* drift;
    const real* ksq;
    real mass_factor, pois_c, fourpi2, inv_n3, scale;
    double* partial;     // [n_streams][ntiles] per-workgroup max|phi|
    int ntiles;
    double* dens;        // optional [n3]: += sum over streams |psi|^2
};

// x pass, forward, from a stored psi: rho = M |psi|^2 of both streams of the pair
// (calculate_density, simulation_object.rs:1031-1063).
template <int N>
__global__ __launch_bounds__((Geom<N, 0>::THREADS)) MSM_OCC(N, 0) void k_rho_pass(PoisArgs a) {
    using G = Geom<N, 0>;
    constexpr int E = G::E, TPL = G::TPL;
    __shared__ cplx lds[G::LDS_ELEMS];
    const int p = blockIdx.y;
    const int sa = a.s0 + 2 * p, sb = sa + 1;
    const bool has_b = (2 * p + 1 < a.nb);
    if (!(a.active[sa] || (has_b && a.active[sb]))) return;

    const Tile<N, 0> t(blockIdx.x, threadIdx.x);
    cplx* tw = lds + G::LPW * G::LINE_LDS;
    load_twiddles<N, 0>(tw, a.tw);
    cplx* line = lds + t.ln * G::LINE_LDS;

    const cplx* __restrict__ xa = a.X + (size_t)(a.x_local ? 2 * p : sa) * a.n3;
    const cplx* __restrict__ xb = a.X + (size_t)(a.x_local ? 2 * p + 1 : sb) * a.n3;
    cplx* __restrict__ w = a.W + (size_t)p * a.n3;

    cplx v[E];
    if (has_b) {
        cplx va[E], vb[E];
#pragma unroll
        for (int q = 0; q < E; ++q) {
            va[q] = xa[t.base + t.j0 + TPL * q];
            vb[q] = xb[t.base + t.j0 + TPL * q];
        }
#pragma unroll
        for (int q = 0; q < E; ++q)
            v[q] = mk<real>(a.mass_factor * (va[q].x * va[q].x + va[q].y * va[q].y),
                            a.mass_factor * (vb[q].x * vb[q].x + vb[q].y * vb[q].y));
    } else {
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const cplx va = xa[t.base + t.j0 + TPL * q];
            v[q] = mk<real>(a.mass_factor * (va.x * va.x + va.y * va.y), (real)0);
        }
    }
    LineFft<real, N, -1, true>::run(v, t.j0, line, tw);
#pragma unroll
    for (int q = 0; q < E; ++q) w[t.base + t.j0 + TPL * q] = v[q];
}

// Last pass of psi = F^-1[...] (along x) fused with the first pass of the Poisson solve:
// for every pair of the batch, finish psi_a and psi_b (drift factor, inverse x transform, unitary
// scale), optionally store them, form rho_a + i rho_b in registers and transform it along x into W.
// One workgroup owns one tile of lines for ALL pairs, so the optional ensemble density
// sum_s |psi_s|^2 is accumulated in registers and added to `dens` without atomics.
template <int N, int DRIFT, int STOREPSI, int DENS>
__global__ __launch_bounds__((Geom<N, 0>::THREADS)) void k_invx_rho(PoisArgs a) {
    using G = Geom<N, 0>;
    constexpr int E = G::E, TPL = G::TPL;
    __shared__ cplx lds[G::LDS_ELEMS];

    const Tile<N, 0> t(blockIdx.x, threadIdx.x);
    cplx* tw = lds + G::LPW * G::LINE_LDS;
    load_twiddles<N, 0>(tw, a.tw);
    cplx* line = lds + t.ln * G::LINE_LDS;

    double dacc[DENS ? E : 1];
    if constexpr (DENS) {
#pragma unroll
        for (int q = 0; q < E; ++q) dacc[q] = 0.0;
    }
    const int pairs = (a.nb + 1) / 2;
    for (int p = 0; p < pairs; ++p) {
        const int sa = a.s0 + 2 * p, sb = sa + 1;
        const bool act_a = a.active[sa] != 0;
        const bool act_b = (2 * p + 1 < a.nb) && a.active[sb] != 0;
        if (!act_a && !act_b) continue;   // uniform over the workgroup
        real rho[2][E];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool act = h == 0 ? act_a : act_b;
            if (act) {
                const int sg = h == 0 ? sa : sb;
                const cplx* __restrict__ x = a.X + (size_t)(2 * p + h) * a.n3;
                cplx v[E];
#pragma unroll
                for (int q = 0; q < E; ++q) v[q] = x[t.base + t.j0 + TPL * q];
                if constexpr (DRIFT) {
                    const cplx* __restrict__ d = a.drift + (size_t)sg * N;
#pragma unroll
                    for (int q = 0; q < E; ++q) v[q] = cmul(v[q], d[t.j0 + TPL * q]);
                }
                LineFft<real, N, +1, true>::run(v, t.j0, line, tw);
                cplx* __restrict__ xo = a.Xout + (size_t)(2 * p + h) * a.n3;
#pragma unroll
                for (int q = 0; q < E; ++q) {
                    v[q] = cscale(v[q], a.scale);
                    if constexpr (STOREPSI) xo[t.base + t.j0 + TPL * q] = v[q];
                    const real m2 = v[q].x * v[q].x + v[q].y * v[q].y;
                    rho[h][q] = a.mass_factor * m2;
                    if constexpr (DENS) dacc[q] += (double)m2;
                }
                if constexpr (G::USES_LDS) exch_sync<true>();   // exchange reads done before the next run writes
            } else {
#pragma unroll
                for (int q = 0; q < E; ++q) rho[h][q] = (real)0;
            }
        }
        cplx v[E];
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = mk<real>(rho[0][q], rho[1][q]);
        LineFft<real, N, -1, true>::run(v, t.j0, line, tw);
        cplx* __restrict__ w = a.W + (size_t)p * a.n3;
#pragma unroll
        for (int q = 0; q < E; ++q) w[t.base + t.j0 + TPL * q] = v[q];
        if constexpr (G::USES_LDS) exch_sync<true>();
    }
    if constexpr (DENS) {
#pragma unroll
        for (int q = 0; q < E; ++q) a.dens[t.base + t.j0 + TPL * q] += dacc[q];
    }
}

// Kick pass, fused with the last pass of the Poisson solve and the first pass of psi_k = F[psi]:
// per pair, phi_a + i phi_b = F_x^-1[W] stays in registers; then for each stream the half-drifted
// psi is rebuilt from the y-transformed scratch exactly as k_invx_rho formed it (drift factor,
// inverse x, unitary scale), kicked psi *= exp(i kick phi) (simulation_object.rs:535-545),
// transformed forward along x and stored with the second drift's x factor (:562-574).
// Neither phi nor the half-drifted psi ever goes to HBM.
template <int N>
__global__ __launch_bounds__((Geom<N, 0>::THREADS)) void k_kick_fwdx(PoisArgs a, const real* __restrict__ kick) {
    using G = Geom<N, 0>;
    constexpr int E = G::E, TPL = G::TPL;
    __shared__ cplx lds[G::LDS_ELEMS];
    const int p = blockIdx.y;
    const int sa = a.s0 + 2 * p, sb = sa + 1;
    const bool act_a = a.active[sa] != 0;
    const bool act_b = (2 * p + 1 < a.nb) && a.active[sb] != 0;
    if (!act_a && !act_b) return;

    const Tile<N, 0> t(blockIdx.x, threadIdx.x);
    cplx* tw = lds + G::LPW * G::LINE_LDS;
    load_twiddles<N, 0>(tw, a.tw);
    cplx* line = lds + t.ln * G::LINE_LDS;

    if constexpr (sizeof(real) == 4) {
        // Single precision has the registers to let the loads run ahead of their use (one 8-wave workgroup per CU would
        // otherwise wait out every load): psi_a is fetched together with W, psi_b before psi_a is worked on.  A finished
        // (or absent) stream of the pair is still transformed -- its scratch slot exists -- and only its store is skipped,
        // so that the code is one straight line.
        const int sbs = (2 * p + 1 < a.nb) ? sb : sa;               // tables of an absent partner: any valid row
        cplx* __restrict__ xa = a.Xout + (size_t)(2 * p) * a.n3;
        cplx* __restrict__ xb = a.Xout + (size_t)(2 * p + 1) * a.n3;
        const cplx* __restrict__ da = a.drift + (size_t)sa * N;
        const cplx* __restrict__ db = a.drift + (size_t)sbs * N;
        const real kca = kick[sa], kcb = kick[sbs];
        real phi[2][E];
        cplx va[E], vb[E];
        {
            const cplx* __restrict__ w = a.W + (size_t)p * a.n3;
            cplx vw[E];
#pragma unroll
            for (int q = 0; q < E; ++q) vw[q] = w[t.base + t.j0 + TPL * q];
#pragma unroll
            for (int q = 0; q < E; ++q) va[q] = xa[t.base + t.j0 + TPL * q];
            LineFft<real, N, +1, true>::run(vw, t.j0, line, tw);
#pragma unroll
            for (int q = 0; q < E; ++q) { phi[0][q] = vw[q].x; phi[1][q] = vw[q].y; }
        }
        auto one = [&](cplx (&v)[E], const cplx* __restrict__ d, real kc, const real (&ph)[E]) {
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = cmul(v[q], d[t.j0 + TPL * q]);
            if constexpr (G::USES_LDS) exch_sync<true>();
            LineFft<real, N, +1, true>::run(v, t.j0, line, tw);
#pragma unroll
            for (int q = 0; q < E; ++q) {
                real sn, cs;
                sincos_t(kc * ph[q], &sn, &cs);
                v[q] = cmul(cscale(v[q], a.scale), mk<real>(cs, sn));
            }
            if constexpr (G::USES_LDS) exch_sync<true>();
            LineFft<real, N, -1, true>::run(v, t.j0, line, tw);
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = cmul(v[q], d[t.j0 + TPL * q]);
        };
#pragma unroll
        for (int q = 0; q < E; ++q) vb[q] = xb[t.base + t.j0 + TPL * q];
        one(va, da, kca, phi[0]);
        if (act_a) {
#pragma unroll
            for (int q = 0; q < E; ++q) xa[t.base + t.j0 + TPL * q] = va[q];
        }
        one(vb, db, kcb, phi[1]);
        if (act_b) {
#pragma unroll
            for (int q = 0; q < E; ++q) xb[t.base + t.j0 + TPL * q] = vb[q];
        }
    } else {
        real phi[2][E];
        {
            const cplx* __restrict__ w = a.W + (size_t)p * a.n3;
            cplx vw[E];
#pragma unroll
            for (int q = 0; q < E; ++q) vw[q] = w[t.base + t.j0 + TPL * q];
            LineFft<real, N, +1, true>::run(vw, t.j0, line, tw);
#pragma unroll
            for (int q = 0; q < E; ++q) { phi[0][q] = vw[q].x; phi[1][q] = vw[q].y; }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const bool act = h == 0 ? act_a : act_b;
            if (!act) continue;   // uniform over the workgroup
            const int sg = h == 0 ? sa : sb;
            cplx* __restrict__ x = a.Xout + (size_t)(2 * p + h) * a.n3;
            const cplx* __restrict__ d = a.drift + (size_t)sg * N;
            const real kc = kick[sg];
            if constexpr (G::USES_LDS) exch_sync<true>();   // also keeps the psi loads below the previous transform
            cplx v[E];
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = x[t.base + t.j0 + TPL * q];
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = cmul(v[q], d[t.j0 + TPL * q]);
            LineFft<real, N, +1, true>::run(v, t.j0, line, tw);
#pragma unroll
            for (int q = 0; q < E; ++q) {
                real sn, cs;
                sincos_t(kc * phi[h][q], &sn, &cs);
                v[q] = cmul(cscale(v[q], a.scale), mk<real>(cs, sn));
            }
            if constexpr (G::USES_LDS) exch_sync<true>();
            LineFft<real, N, -1, true>::run(v, t.j0, line, tw);
#pragma unroll
            for (int q = 0; q < E; ++q) x[t.base + t.j0 + TPL * q] = cmul(v[q], d[t.j0 + TPL * q]);
        }
    }
}

// z pass, fused: forward FFT, Green's multiply -C / k^2 with the k = 0 mode
// zeroed (simulation_object.rs:1076-1102), inverse FFT.  The unitary scales of
// both 3-D transforms (1/N^3 in total, exact) ride on the Green's factor.
template <int N>
__global__ __launch_bounds__((Geom<N, 2>::THREADS)) MSM_OCC(N, 2) void k_poisson_mid(PoisArgs a) {
    using G = Geom<N, 2>;
    constexpr int E = G::E, TPL = G::TPL;
    __shared__ cplx lds[G::LDS_ELEMS];
    const int p = blockIdx.y;
    const int sa = a.s0 + 2 * p, sb = sa + 1;
    if (!(a.active[sa] || (2 * p + 1 < a.nb && a.active[sb]))) return;

    const Tile<N, 2> t(blockIdx.x, threadIdx.x);
    cplx* tw = lds + G::LPW * G::LINE_LDS;
    load_twiddles<N, 2>(tw, a.tw);
    // k_i^2 through LDS: sixteen global reads per thread would be hoisted above the transform and held in registers
    __shared__ real kq[N];
    for (int i = threadIdx.x; i < N; i += G::THREADS) kq[i] = a.ksq[i];
    if constexpr (!G::USES_LDS) lds_barrier();   // otherwise the exchange barriers of the forward run lie in between
    cplx* line = lds + t.ln * G::LINE_LDS;
    cplx* __restrict__ w = a.W + (size_t)p * a.n3;

    cplx v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) v[q] = w[t.base + (size_t)(t.j0 + TPL * q) * t.stride];
    LineFft<real, N, -1>::run(v, t.j0, line, tw);

    const real kxy = kq[t.c1] + kq[t.c2];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const real k2 = (kxy + kq[t.j0 + TPL * q]) * a.fourpi2;
        const real g = (k2 == (real)0) ? (real)0 : (a.pois_c * recip_t(k2)) * a.inv_n3;
        v[q] = cscale(v[q], g);
    }
    if constexpr (G::USES_LDS) lds_barrier();  // all exchange reads of the forward run are done
    LineFft<real, N, +1>::run(v, t.j0, line, tw);
    // the store addresses are rebuilt here: kept alive since the loads they would cost 32 registers
    const size_t sbase = (size_t)fresh((unsigned)t.base) + (size_t)fresh((unsigned)t.j0) * t.stride;
#pragma unroll
    for (int q = 0; q < E; ++q) w[sbase + (size_t)(TPL * q) * t.stride] = v[q];
}

// x pass, inverse: split phi_a / phi_b (real(phi), simulation_object.rs:1109) and/or
// record this workgroup's max|phi| per stream (max_all(abs(phi)), :905).
template <int N, int STOREPHI, int TRACKMAX>
__global__ __launch_bounds__((Geom<N, 0>::THREADS)) MSM_OCC(N, 0) void k_phi_out(PoisArgs a) {
    using G = Geom<N, 0>;
    constexpr int E = G::E, TPL = G::TPL;
    __shared__ cplx lds[G::LDS_ELEMS];
    __shared__ double red[TRACKMAX ? 32 : 1];
    const int p = blockIdx.y;
    const int sa = a.s0 + 2 * p, sb = sa + 1;
    const bool has_b = (2 * p + 1 < a.nb);
    if (!(a.active[sa] || (has_b && a.active[sb]))) return;

    const Tile<N, 0> t(blockIdx.x, threadIdx.x);
    cplx* tw = lds + G::LPW * G::LINE_LDS;
    load_twiddles<N, 0>(tw, a.tw);
    cplx* line = lds + t.ln * G::LINE_LDS;
    const cplx* __restrict__ w = a.W + (size_t)p * a.n3;

    cplx v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) v[q] = w[t.base + t.j0 + TPL * q];
    LineFft<real, N, +1, true>::run(v, t.j0, line, tw);

    real* __restrict__ pa = a.PHI + (size_t)(2 * p) * a.n3;
    real* __restrict__ pb = a.PHI + (size_t)(2 * p + 1) * a.n3;
    double ma = 0.0, mb = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const size_t g = t.base + t.j0 + TPL * q;
        if constexpr (STOREPHI) {
            pa[g] = v[q].x;
            if (has_b) pb[g] = v[q].y;
        }
        if constexpr (TRACKMAX) {
            const double ax = fabs((double)v[q].x), ay = fabs((double)v[q].y);
            ma = (ax != ax) ? ax : ((ma != ma) ? ma : (ax > ma ? ax : ma));
            mb = (ay != ay) ? ay : ((mb != mb) ? mb : (ay > mb ? ay : mb));
        }
    }
    if constexpr (TRACKMAX) {
        block_max2<G::THREADS>(ma, mb, red);
        if (threadIdx.x == 0) {
            a.partial[(size_t)sa * a.ntiles + blockIdx.x] = ma;
            if (has_b) a.partial[(size_t)sb * a.ntiles + blockIdx.x] = mb;
        }
    }
}

// ---------------------------------------------------------------------------
// Two-pass Poisson solve for N = 256 (y split as 16 x 16):
//   pass A  (x lines and the outer radix-16 stage of y) lives inside the x-pass kernels of psi,
//   pass B  (inner radix-16 stage of y, all of z, Green's multiply, and back) is one kernel,
// so the pair-packed work array crosses HBM twice per solve instead of four times.
// With y = b + 16 t, ky = t' + 16 b':  F_y = DFT16_b[ w^(b t') DFT16_t[.] ],  w = e^(-2 pi i / 256).
// An A tile is 32 x-lines: 16 rows y = b + 16 t of the two planes z = 2 zh + zl.  Between the passes
// a grid is stored as
//     T[t'][kx >> 4][b][zh][(kx >> 2) & 3][kx bit 1][zl][kx bit 0]
// The innermost 8 elements (128 B) are what an A tile and a B run have in common.  Four such blocks
// with neighbouring kx sit side by side: an A tile (fixed b, zh; all t', kx) moves 512-byte pieces,
// a B run (fixed t', kx >> 2; all b, z) moves 128-byte pieces 512 B apart whose neighbours belong to
// the next three runs, i.e. to workgroups in flight at the same time.  Measured as plain copies this
// split serves both sides at 5.0 / 5.7 TB/s; fully contiguous B runs left the A side at 3.8 TB/s.
// ---------------------------------------------------------------------------
namespace tp {
constexpr int N = 256;
constexpr int A_THREADS = 512, A_LINE = Geom<N, 0>::LINE_LDS;
constexpr int A_HALF = 16 * A_LINE + 8;            // zl = 1 lines start 128 B off the bank period of zl = 0
constexpr int A_LDS = 2 * A_HALF + N;              // + twiddle table
constexpr int A_TILES = 2048;                      // zh (128) x b (16)
constexpr size_t T_TP = (size_t)64 * 16384;        // stride of t' in T (elements)
constexpr int B_THREADS = 1024;                    // one workgroup per run (t', kx >> 2): 16 b x 256 z x 4 kx
constexpr int B_LINE = 272;                        // reals per padded z line; 272 = 16 mod 32 keeps line pairs on disjoint banks
constexpr int B_LINES = 64;                        // z lines per workgroup: 16 b' x 4 kx
constexpr int B_PLANE = B_LINES * B_LINE + 8;
constexpr int B_BLOCKS = 1024;                     // runs per grid
constexpr int KL = 4;                              // 128-byte blocks of neighbouring kx side by side in T
constexpr size_t T_B = (size_t)128 * KL * 8;       // stride of b in T (elements)
constexpr int MAX_BATCH = 256;                     // streams per sub-batch the A kernels keep flags for
}

struct TileA {
    int j0, zl, t, b, zh;
    size_t base;     // first element of this thread's x line in the natural layout
    cplx* line;      // its padded LDS line
    __device__ __forceinline__ TileA(int tile, int tid, cplx* lds) {
        j0 = tid & 15;
        const int ln = tid >> 4;
        zl = ln >> 4; t = ln & 15;
        zh = tile >> 4; b = tile & 15;
        base = ((size_t)(2 * zh + zl) * tp::N + (b + 16 * t)) * tp::N;
        line = lds + zl * tp::A_HALF + t * tp::A_LINE;
    }
};

// Line role (thread = 16 kx of one line) -> column role (thread = one (kx, zl), the 16 lines t),
// radix-16 stage over t, store to T.  Barriers inside; all 512 threads call.
__device__ __forceinline__ void a_store_T(cplx (&v)[16], const TileA& t, cplx* lds, cplx* __restrict__ wT, size_t grid_off) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 16; ++q) t.line[lds_pad(t.j0 + 16 * q)] = v[q];   // own line: ordered within the wave
    lds_barrier();
    const int kxh = tid >> 3, zl = (tid >> 1) & 1, kx = (kxh << 2) | (((tid >> 2) & 1) << 1) | (tid & 1);
    const cplx* col = lds + zl * tp::A_HALF + lds_pad(kx);
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = col[q * tp::A_LINE];
    dft<16, -1, real>(v);
    const unsigned off = (unsigned)(((((kxh / tp::KL) * 16 + t.b) * 128 + t.zh) * tp::KL + (kxh % tp::KL)) * 8 + (tid & 7));
#pragma unroll
    for (int q = 0; q < 16; ++q) (wT + (grid_off + (size_t)q * tp::T_TP))[off] = v[q];
}

// The way back, in two halves so that the loads can be issued ahead of their use: a_fetch_T only
// loads (column role); a_cols_to_lines does the inverse radix-16 stage over t' and the column role ->
// line role exchange (one barrier; LDS lines must not be in use by other wavefronts on entry).
// The sixteen loads lie 16 MiB apart; one running address pair serves them all.
typedef real real2v __attribute__((ext_vector_type(2)));
typedef const __attribute__((address_space(1))) real2v* gcptr;
__device__ __forceinline__ void a_fetch_T(cplx (&v)[16], const TileA& t, const cplx* __restrict__ wT, size_t grid_off) {
    const int tid = threadIdx.x;
    const int kxh = tid >> 3;
    gcptr p = (gcptr)(wT + grid_off + (size_t)(((((kxh / tp::KL) * 16 + t.b) * 128 + t.zh) * tp::KL + (kxh % tp::KL)) * 8 + (tid & 7)));
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const real2v d = *p;
        v[q] = mk<real>(d.x, d.y);
        p += tp::T_TP;
        asm volatile("" : "+v"(p));
    }
}
__device__ __forceinline__ void a_cols_to_lines(cplx (&v)[16], const TileA& t, cplx* lds) {
    const int tid = threadIdx.x;
    const int kxh = tid >> 3, zl = (tid >> 1) & 1, kx = (kxh << 2) | (((tid >> 2) & 1) << 1) | (tid & 1);
    dft<16, +1, real>(v);
    cplx* col = lds + zl * tp::A_HALF + lds_pad(kx);
#pragma unroll
    for (int q = 0; q < 16; ++q) col[q * tp::A_LINE] = v[q];
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = t.line[lds_pad(t.j0 + 16 * q)];
}
__device__ __forceinline__ void a_load_T(cplx (&v)[16], const TileA& t, cplx* lds, const cplx* __restrict__ wT, size_t grid_off) {
    a_fetch_T(v, t, wT, grid_off);
    a_cols_to_lines(v, t, lds);
}

__device__ __forceinline__ void a_load_twiddles(cplx* tw_lds, const cplx* __restrict__ tw_g) {
    if (threadIdx.x < tp::N) tw_lds[threadIdx.x] = tw_g[TwLayout<tp::N>::source(threadIdx.x)];
    lds_barrier();
}

// rho from a stored psi on A tiles (first step, msm_potential).
__global__ __launch_bounds__(tp::A_THREADS) void k_rho_pass2(PoisArgs a) {
    constexpr int N = tp::N, E = 16, TPL = 16;
    __shared__ cplx lds[tp::A_LDS];
    const int p = blockIdx.y;
    const int sa = a.s0 + 2 * p, sb = sa + 1;
    const bool has_b = (2 * p + 1 < a.nb);
    if (!(a.active[sa] || (has_b && a.active[sb]))) return;
    const TileA t(blockIdx.x, threadIdx.x, lds);
    cplx* tw = lds + 2 * tp::A_HALF;
    a_load_twiddles(tw, a.tw);
    const cplx* __restrict__ xa = a.X + (size_t)(a.x_local ? 2 * p : sa) * a.n3;
    const cplx* __restrict__ xb = a.X + (size_t)(a.x_local ? 2 * p + 1 : sb) * a.n3;
    cplx v[E];
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const cplx va = xa[t.base + t.j0 + TPL * q];
        real rb = (real)0;
        if (has_b) {
            const cplx vb = xb[t.base + t.j0 + TPL * q];
            rb = a.mass_factor * (vb.x * vb.x + vb.y * vb.y);
        }
        v[q] = mk<real>(a.mass_factor * (va.x * va.x + va.y * va.y), rb);
    }
    LineFft<real, N, -1, true>::run(v, t.j0, t.line, tw);
    a_store_T(v, t, lds, a.W, (size_t)p * a.n3);
}

// k_phi_out on A tiles.
template <int STOREPHI, int TRACKMAX>
__global__ __launch_bounds__(tp::A_THREADS) void k_phi_out2(PoisArgs a) {
    constexpr int N = tp::N, E = 16, TPL = 16;
    __shared__ cplx lds[tp::A_LDS];
    __shared__ double red[TRACKMAX ? 32 : 1];
    const int p = blockIdx.y;
    const int sa = a.s0 + 2 * p, sb = sa + 1;
    const bool has_b = (2 * p + 1 < a.nb);
    if (!(a.active[sa] || (has_b && a.active[sb]))) return;
    const TileA t(blockIdx.x, threadIdx.x, lds);
    cplx* tw = lds + 2 * tp::A_HALF;
    a_load_twiddles(tw, a.tw);
    cplx v[E];
    a_load_T(v, t, lds, a.W, (size_t)p * a.n3);
    LineFft<real, N, +1, true>::run(v, t.j0, t.line, tw);

    real* __restrict__ pa = a.PHI + (size_t)(2 * p) * a.n3;
    real* __restrict__ pb = a.PHI + (size_t)(2 * p + 1) * a.n3;
    double ma = 0.0, mb = 0.0;
#pragma unroll
    for (int q = 0; q < E; ++q) {
        const size_t g = t.base + t.j0 + TPL * q;
        if constexpr (STOREPHI) {
            pa[g] = v[q].x;
            if (has_b) pb[g] = v[q].y;
        }
        if constexpr (TRACKMAX) {
            ma = nanmax(fabs((double)v[q].x), ma);
            mb = nanmax(fabs((double)v[q].y), mb);
        }
    }
    if constexpr (TRACKMAX) {
        block_max2<tp::A_THREADS>(ma, mb, red);
        if (threadIdx.x == 0) {
            a.partial[(size_t)sa * a.ntiles + blockIdx.x] = ma;
            if (has_b) a.partial[(size_t)sb * a.ntiles + blockIdx.x] = mb;
        }
    }
}

// max|phi| only (the potential that sizes the next timestep, simulation_object.rs:497,905): one workgroup
// owns a tile for all pairs of the batch and fetches the next pair's tile while it transforms the
// current one (straight-line body, stream flags in LDS: see k_invx_rho3).
__global__ __launch_bounds__(tp::A_THREADS) void k_phi_max3(PoisArgs a) {
    constexpr int N = tp::N, E = 16;
    __shared__ cplx lds[tp::A_LDS];
    __shared__ double red[32];
    __shared__ int act_lds[tp::MAX_BATCH + 2];
    const TileA t(blockIdx.x, threadIdx.x, lds);
    cplx* tw = lds + 2 * tp::A_HALF;
    for (int i = threadIdx.x; i < tp::MAX_BATCH + 2; i += tp::A_THREADS) act_lds[i] = (i < a.nb) ? a.active[a.s0 + i] : 0;
    a_load_twiddles(tw, a.tw);   // barrier inside
    const int pairs = (a.nb + 1) / 2;
    cplx va[E], vb[E];
    a_fetch_T(va, t, a.W, 0);
    auto one = [&](cplx (&v)[E], int p) {
        const bool act_a = act_lds[2 * p] != 0, act_b = act_lds[2 * p + 1] != 0;
        a_cols_to_lines(v, t, lds);
        LineFft<real, N, +1, true>::run(v, t.j0, t.line, tw);
        double ma = 0.0, mb = 0.0;
#pragma unroll
        for (int q = 0; q < E; ++q) {
            ma = nanmax(fabs((double)v[q].x), ma);
            mb = nanmax(fabs((double)v[q].y), mb);
        }
        block_max2<tp::A_THREADS>(ma, mb, red);   // its barrier also separates this exchange from the next column writes
        if (threadIdx.x == 0) {
            if (act_a) a.partial[(size_t)(a.s0 + 2 * p) * a.ntiles + blockIdx.x] = ma;
            if (act_b) a.partial[(size_t)(a.s0 + 2 * p + 1) * a.ntiles + blockIdx.x] = mb;
        }
        lds_barrier();   // red[] is reused by the next pair
    };
    for (int p = 0; p < pairs; p += 2) {
        a_fetch_T(vb, t, a.W, (size_t)min(p + 1, pairs - 1) * a.n3);
        one(va, p);
        a_fetch_T(va, t, a.W, (size_t)min(p + 2, pairs - 1) * a.n3);
        if (p + 1 < pairs) one(vb, p + 1);
    }
}

// Pass-B role changes through one real LDS plane (32 padded z lines).
//   column role: thread = (z = tid >> 1, kx bit 0 = tid & 1), holds the 16 values b (or b')
//   line role:   thread = (j0 = tid & 15, line L = tid >> 4 = 2 b' + kx bit 0), holds z = j0 + 16 q
struct RolesB {
    int j0, L, bprime, kxl;   // line role: z = j0 + 16 q of line L = (b', kx low bits kxl)
    real* colw;
    real* line;
    // Lines are stored as L = (kx bit 1) * 32 + 2 b' + (kx bit 0); the upper half is skewed by 8 reals
    // so that the four kx of a column access fall on distinct bank quarters.
    static __device__ __forceinline__ int line_off(int L) { return L * tp::B_LINE + (L >> 5) * 8; }
    __device__ __forceinline__ RolesB(real* plane) {
        const int tid = threadIdx.x;
        j0 = tid & 15; L = tid >> 4;
        line = plane + line_off(L);
        bprime = (L >> 1) & 15; kxl = ((L >> 5) << 1) | (L & 1);
        // column role: the lane's position in the 8-element block is (kx bit 1, zl, kx bit 0), zh = tid >> 3
        const int inner = tid & 7, z = 2 * (tid >> 3) + ((inner >> 1) & 1);
        colw = plane + line_off((inner >> 2) * 32 + (inner & 1)) + lds_pad(z);
    }
};
// column -> line; the plane must be free on entry and is free again on return
__device__ __forceinline__ void b_to_lines(cplx (&v)[16], const RolesB& r) {
    constexpr int CS = 2 * tp::B_LINE;
#pragma unroll
    for (int q = 0; q < 16; ++q) r.colw[q * CS] = v[q].x;
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].x = r.line[lds_pad(r.j0 + 16 * q)];
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) r.colw[q * CS] = v[q].y;
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].y = r.line[lds_pad(r.j0 + 16 * q)];
}
// line -> column; the plane must be free on entry; other threads may still read it on return
__device__ __forceinline__ void b_to_cols(cplx (&v)[16], const RolesB& r) {
    constexpr int CS = 2 * tp::B_LINE;
#pragma unroll
    for (int q = 0; q < 16; ++q) r.line[lds_pad(r.j0 + 16 * q)] = v[q].x;
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].x = r.colw[q * CS];
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) r.line[lds_pad(r.j0 + 16 * q)] = v[q].y;
    lds_barrier();
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q].y = r.colw[q * CS];
}
// blockIdx.x = run = (t', kx >> 2); neighbouring runs interleave in memory and run side by side
struct RunB {
    int run, tpi, kxh;
    size_t off;       // first element of this run within a grid
    unsigned toff;    // this lane's element within it (column role): (zh, position in the block)
    __device__ __forceinline__ RunB() {
        const int tid = threadIdx.x;
        run = blockIdx.x; tpi = run >> 6; kxh = run & 63;
        off = (size_t)(tpi * (64 / tp::KL) + kxh / tp::KL) * 16 * tp::T_B + (kxh % tp::KL) * 8;
        toff = (unsigned)((tid >> 3) * (tp::KL * 8) + (tid & 7));
    }
    __device__ __forceinline__ int kx(const RolesB& r) const { return (kxh << 2) | r.kxl; }
    __device__ __forceinline__ int ky(const RolesB& r) const { return tpi + 16 * r.bprime; }
};
// T (column role) -> inner y stage -> line role -> z transform; v ends as kz = j0 + 16 q
__device__ __forceinline__ void b_forward(cplx (&v)[16], const cplx* __restrict__ T, size_t trun, unsigned toff,
                                          int tpi, const RolesB& r, const cplx* tw_lds, const cplx* __restrict__ tw_g) {
    const bufdesc rs = buf_of(T + trun);
#pragma unroll
    for (int q = 0; q < 16; ++q) buf_ld(v[q], rs, toff * (unsigned)sizeof(cplx), (unsigned)(q * tp::T_B * sizeof(cplx)));
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmul(v[q], tw_g[(q * tpi) & (tp::N - 1)]);
    dft<16, -1, real>(v);
    b_to_lines(v, r);
    LineFftSplit<real, tp::N, -1, true>::run(v, r.j0, r.line, tw_lds);
}
// the way back; the plane must be free on entry
__device__ __forceinline__ void b_inverse(cplx (&v)[16], cplx* __restrict__ T, size_t trun, unsigned toff, int tpi,
                                          const RolesB& r, const cplx* tw_lds, const cplx* __restrict__ tw_g) {
    LineFftSplit<real, tp::N, +1, true>::run(v, r.j0, r.line, tw_lds);
    b_to_cols(v, r);
    dft<16, +1, real>(v);
#pragma unroll
    for (int q = 1; q < 16; ++q) v[q] = cmulc(v[q], tw_g[(q * tpi) & (tp::N - 1)]);
    const bufdesc rs = buf_of(T + trun);
#pragma unroll
    for (int q = 0; q < 16; ++q) buf_st(v[q], rs, toff * (unsigned)sizeof(cplx), (unsigned)(q * tp::T_B * sizeof(cplx)));
}

// Pass B of the Poisson solve: inner y stage, z transform, Green's multiply -C / k^2 (k = 0 zeroed,
// simulation_object.rs:1076-1102), and both stages back, on one run of T in place.
__global__ __launch_bounds__(tp::B_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_poisson_b(PoisArgs a) {
    constexpr int N = tp::N;
    __shared__ real plane[tp::B_PLANE];
    __shared__ cplx tw[N];
    const int tid = threadIdx.x;
    const int p = blockIdx.y;
    const int sa = a.s0 + 2 * p, sb = sa + 1;
    if (!(a.active[sa] || (2 * p + 1 < a.nb && a.active[sb]))) return;
    __shared__ real kq[N];
    if (tid < N) tw[tid] = a.tw[TwLayout<N>::source(tid)];   // first reads are behind the exchange barriers
    else if (tid < 2 * N) kq[tid - N] = a.ksq[tid - N];
    const RunB rb;
    const RolesB r(plane);
    const size_t trun = (size_t)p * a.n3 + rb.off;
    const unsigned toff = rb.toff;

    cplx v[16];
    b_forward(v, a.W, trun, toff, rb.tpi, r, tw, a.tw);
    const int kx = rb.kx(r), ky = rb.ky(r);
    const real kxy = kq[kx] + kq[ky];   // ((kx^2 + ky^2) + kz^2) * (2 pi)^2, utils/fft.rs:140-160
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const real k2 = (kxy + kq[r.j0 + 16 * q]) * a.fourpi2;
        const real g = (k2 == (real)0) ? (real)0 : (a.pois_c * recip_t(k2)) * a.inv_n3;
        v[q] = cscale(v[q], g);
    }
    b_inverse(v, a.W, trun, toff, rb.tpi, r, tw, a.tw);
}

// Pass B of the psi transforms.  Between steps a stream's state is S = pass A of psi (psi after the forward x
// lines and the outer y stage, in the T layout): psi_k = B[S] / N^{3/2} and psi = A^-1[S] / N^{3/2}, so psi_k
// itself never has to be stored.  Both kinetic half-steps are then the same round trip through Fourier space,
//   T -> pass B -> x D / N^{3/2} [-> aliased-mass partial] -> pass B^-1 -> T'
// (first half: S -> scratch, simulation_object.rs:504-528; second half: kicked psi -> S, :552-581 with
// check_alias :1249-1280), 16 B in and 16 B out per cell each.
struct PsiBArgs {
    const cplx* src;     // [.][n3] in the T layout, indexed by blockIdx.y
    cplx* dst;           // [.][n3], indexed by blockIdx.y (T layout; k_psi_b_fwd: psi_k as [kx][ky][kz])
    size_t n3;
    const int* active;   // may be null
    int s0;
    const cplx* tw;
    const cplx* drift;   // [n_streams][N]
    const real* ksq;
    real scale, alias_cut, fourpi2;
    double* partial;     // [n_streams][ntiles]
    int ntiles;
};

template <int ALIAS>
__global__ __launch_bounds__(tp::B_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_psi_rt(PsiBArgs a) {
    constexpr int N = tp::N;
    __shared__ real plane[tp::B_PLANE];
    __shared__ cplx tw[N];
    __shared__ double red[ALIAS ? tp::B_THREADS / 64 : 1];
    __shared__ cplx dl[N];                   // this stream's 1-D drift factors
    __shared__ real kq[ALIAS ? N : 1];       // k_i^2 table
    const int tid = threadIdx.x;
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (a.active != nullptr && !a.active[sg]) return;
    // tables go to LDS up front: read from global memory in the middle of the kernel they would either expose an L2
    // round trip behind the transform or be hoisted into sixteen live register pairs
    if (tid < N) tw[tid] = a.tw[TwLayout<N>::source(tid)];
    else if (tid < 2 * N) dl[tid - N] = a.drift[(size_t)sg * N + (tid - N)];
    else if (ALIAS && tid < 3 * N) kq[tid - 2 * N] = a.ksq[tid - 2 * N];
    lds_barrier();   // the z lines exchange inside their wavefront: no other barrier lies before the first table read
    const RunB rb;
    const RolesB r(plane);
    const size_t trun = (size_t)ls * a.n3 + rb.off;
    const unsigned toff = rb.toff;

    cplx v[16];
    b_forward(v, a.src, trun, toff, rb.tpi, r, tw, a.tw);
    const int kx = rb.kx(r), ky = rb.ky(r);
    {
        const cplx dxy = cscale(cmul(dl[kx], dl[ky]), a.scale);
        // four factors at a time: all sixteen table reads hoisted to the front would not fit the register budget
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * c + i] = cmul(v[4 * c + i], cmul(dxy, dl[r.j0 + 16 * (4 * c + i)]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if constexpr (ALIAS) {
        const real kxy = kq[kx] + kq[ky];
        double acc = 0.0;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const real k2 = (kxy + kq[r.j0 + 16 * q]) * a.fourpi2;
            if (k2 > a.alias_cut) acc += (double)(v[q].x * v[q].x + v[q].y * v[q].y);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        lds_barrier();
        if (tid == 0) {
            double tot = red[0];
#pragma unroll
            for (int w = 1; w < tp::B_THREADS / 64; ++w) tot += red[w];
            a.partial[(size_t)sg * a.ntiles + blockIdx.x] = tot;
        }
    }
    b_inverse(v, a.dst, trun, toff, rb.tpi, r, tw, a.tw);
}

// psi_k = B[S] / N^{3/2} written as [kx][ky][kz] (kz contiguous, 256-byte rows): downloads of psi_k and the
// synthesizer's psi_k products; k_transpose_xz turns it into the reference's element order.
__global__ __launch_bounds__(tp::B_THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_psi_b_fwd(PsiBArgs a) {
    constexpr int N = tp::N;
    __shared__ real plane[tp::B_PLANE];
    __shared__ cplx tw[N];
    const int tid = threadIdx.x;
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (a.active != nullptr && !a.active[sg]) return;
    if (tid < N) tw[tid] = a.tw[TwLayout<N>::source(tid)];   // first read is behind the role-change barriers
    const RunB rb;
    const RolesB r(plane);
    const int kx = rb.kx(r), ky = rb.ky(r);
    cplx* __restrict__ prow = a.dst + (size_t)ls * a.n3 + ((size_t)kx * N + ky) * N + r.j0;
    cplx v[16];
    b_forward(v, a.src, (size_t)ls * a.n3 + rb.off, rb.toff, rb.tpi, r, tw, a.tw);
#pragma unroll
    for (int q = 0; q < 16; ++q) prow[16 * q] = cscale(v[q], a.scale);
}

// Pass A of a plain psi transform: natural layout <-> T.
//   SGN = -1: x lines forward, outer y stage, store T;   SGN = +1: the inverse, scaled, to the natural layout.
template <int SGN>
__global__ __launch_bounds__(tp::A_THREADS) void k_psi_a(PsiArgs a) {
    constexpr int N = tp::N, E = 16, TPL = 16;
    __shared__ cplx lds[tp::A_LDS];
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (a.active != nullptr && !a.active[sg]) return;
    const size_t so = (size_t)(a.src_local ? ls : sg) * a.n3, dof = (size_t)(a.dst_local ? ls : sg) * a.n3;
    const TileA t(blockIdx.x, threadIdx.x, lds);
    cplx* tw = lds + 2 * tp::A_HALF;
    a_load_twiddles(tw, a.tw);
    cplx v[E];
    if constexpr (SGN < 0) {
#pragma unroll
        for (int q = 0; q < E; ++q) v[q] = a.src[so + t.base + t.j0 + TPL * q];
        LineFft<real, N, -1, true>::run(v, t.j0, t.line, tw);
        a_store_T(v, t, lds, a.dst, dof);
    } else {
        a_load_T(v, t, lds, a.src, so);
        LineFft<real, N, +1, true>::run(v, t.j0, t.line, tw);
#pragma unroll
        for (int q = 0; q < E; ++q) a.dst[dof + t.base + t.j0 + TPL * q] = cscale(v[q], a.scale);
    }
}

// k_invx_rho with psi arriving in T (after pass B of the inverse transform): no drift factor left to apply.
// The workgroup is alone on its CU (LDS), so it hides the load latency itself: stream b of a pair is
// fetched while stream a is transformed, stream a of the next pair while W is.  The loop body is
// straight-line code -- a finished stream is still transformed and its density dropped by a select,
// the fetch past the end re-reads the last tile -- so that the number of loads in flight at every use
// is a compile-time constant: the vector-memory counter retires in order, and a wait placed at a
// branch merge has to assume the shorter path, i.e. drain everything.  For the same reason the stream
// flags are read once into LDS: a flag load between a fetch and its use would wait for the fetch.
template <int DENS>
__global__ __launch_bounds__(tp::A_THREADS) void k_invx_rho3(PoisArgs a) {
    constexpr int E = 16, TPL = 16, N = tp::N;
    __shared__ cplx lds[tp::A_LDS];
    __shared__ int act_lds[tp::MAX_BATCH + 2];
    const TileA t(blockIdx.x, threadIdx.x, lds);
    cplx* tw = lds + 2 * tp::A_HALF;
    for (int i = threadIdx.x; i < tp::MAX_BATCH + 2; i += tp::A_THREADS) act_lds[i] = (i < a.nb) ? a.active[a.s0 + i] : 0;
    a_load_twiddles(tw, a.tw);   // barrier inside

    double dacc[DENS ? E : 1];
    if constexpr (DENS) {
#pragma unroll
        for (int q = 0; q < E; ++q) dacc[q] = 0.0;
    }
    const int pairs = (a.nb + 1) / 2, last = a.nb - 1;
    cplx va[E], vb[E];
    a_fetch_T(va, t, a.X, 0);
    auto pair_body = [&](int p) {
        const bool act_a = act_lds[2 * p] != 0, act_b = act_lds[2 * p + 1] != 0;   // uniform over the workgroup
        a_fetch_T(vb, t, a.X, (size_t)min(2 * p + 1, last) * a.n3);
        real rho0[E];
        a_cols_to_lines(va, t, lds);
        LineFft<real, N, +1, true>::run(va, t.j0, t.line, tw);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const cplx u = cscale(va[q], a.scale);
            const real m2 = u.x * u.x + u.y * u.y;
            rho0[q] = act_a ? a.mass_factor * m2 : (real)0;
            if constexpr (DENS) dacc[q] += act_a ? (double)m2 : 0.0;
        }
        lds_barrier();   // line role -> column writes of the next exchange
        a_cols_to_lines(vb, t, lds);
        LineFft<real, N, +1, true>::run(vb, t.j0, t.line, tw);
#pragma unroll
        for (int q = 0; q < E; ++q) {
            const cplx u = cscale(vb[q], a.scale);
            const real m2 = u.x * u.x + u.y * u.y;
            vb[q] = mk<real>(rho0[q], act_b ? a.mass_factor * m2 : (real)0);
            if constexpr (DENS) dacc[q] += act_b ? (double)m2 : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
        a_fetch_T(va, t, a.X, (size_t)min(2 * p + 2, last) * a.n3);
        LineFft<real, N, -1, true>::run(vb, t.j0, t.line, tw);
        a_store_T(vb, t, lds, a.W, (size_t)p * a.n3);   // a pair of finished streams stores zeros
        lds_barrier();
    };
    // first pair peeled: the loop is then entered with the same loads and stores in flight as its back edge leaves
    pair_body(0);
    for (int p = 1; p < pairs; ++p) pair_body(p);
    if constexpr (DENS) {
#pragma unroll
        for (int q = 0; q < E; ++q) a.dens[t.base + t.j0 + TPL * q] += dacc[q];
    }
}

// k_kick_fwdx with psi in T on both sides: phi_a + i phi_b and the half-drifted psi come back through
// the outer y stage and the inverse x lines, the kicked psi leaves through forward x and the outer y stage.
// Loads run ahead of their use (psi_a with W, psi_b from the middle of psi_a's turn); both streams of
// the pair are always processed (see k_invx_rho3), only the store of a finished stream is skipped.
__global__ __launch_bounds__(tp::A_THREADS) void k_kick3(PoisArgs a, const real* __restrict__ kick) {
    constexpr int N = tp::N, E = 16;
    __shared__ cplx lds[tp::A_LDS];
    const int p = blockIdx.y;
    const int sa = a.s0 + 2 * p;
    const bool has_b = 2 * p + 1 < a.nb;
    const int sb = has_b ? sa + 1 : sa;
    const bool act_a = a.active[sa] != 0;
    const bool act_b = has_b && a.active[sb] != 0;
    if (!act_a && !act_b) return;
    const real kca = kick[sa], kcb = kick[sb];
    const size_t xa = (size_t)(2 * p) * a.n3, xb = (size_t)(has_b ? 2 * p + 1 : 2 * p) * a.n3;
    const TileA t(blockIdx.x, threadIdx.x, lds);
    cplx* tw = lds + 2 * tp::A_HALF;
    a_load_twiddles(tw, a.tw);

    cplx va[E], vb[E];
    real phi[2][E];
    {
        cplx vw[E];
        a_fetch_T(vw, t, a.W, (size_t)p * a.n3);
        a_fetch_T(va, t, a.X, xa);
        a_cols_to_lines(vw, t, lds);
        LineFft<real, N, +1, true>::run(vw, t.j0, t.line, tw);
#pragma unroll
        for (int q = 0; q < E; ++q) { phi[0][q] = vw[q].x; phi[1][q] = vw[q].y; }
        lds_barrier();
    }
    a_cols_to_lines(va, t, lds);
    LineFft<real, N, +1, true>::run(va, t.j0, t.line, tw);
#pragma unroll
    for (int q = 0; q < E; ++q) {
        real sn, cs;
        sincos_t(kca * phi[0][q], &sn, &cs);
        va[q] = cmul(cscale(va[q], a.scale), mk<real>(cs, sn));
    }
    LineFft<real, N, -1, true>::run(va, t.j0, t.line, tw);
    if (act_a) a_store_T(va, t, lds, a.Xout, xa);
    lds_barrier();

    a_fetch_T(vb, t, a.X, xb);
    a_cols_to_lines(vb, t, lds);
    LineFft<real, N, +1, true>::run(vb, t.j0, t.line, tw);
#pragma unroll
    for (int q = 0; q < E; ++q) {
        real sn, cs;
        sincos_t(kcb * phi[1][q], &sn, &cs);
        vb[q] = cmul(cscale(vb[q], a.scale), mk<real>(cs, sn));
    }
    LineFft<real, N, -1, true>::run(vb, t.j0, t.line, tw);
    if (act_b) a_store_T(vb, t, lds, a.Xout, xb);
}

// dst[z][y][x] = src[x][y][z] for one N^3 grid (psi_k downloads from the P[kx][ky][kz] layout)
__global__ __launch_bounds__(256) void k_transpose_xz(const cplx* __restrict__ src, cplx* __restrict__ dst, int N) {
    __shared__ cplx tile[32][33];
    const int y = blockIdx.z, x0 = blockIdx.x * 32, z0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) tile[i][tx] = src[((size_t)(x0 + i) * N + y) * N + z0 + tx];
    __syncthreads();
    for (int i = ty; i < 32; i += 8) dst[((size_t)(z0 + i) * N + y) * N + x0 + tx] = tile[tx][i];
}

// Reduce the per-workgroup partials of one stream in a fixed order (bitwise reproducible).
// MODE 0: sum -> alias_sum;  MODE 1: max (NaN wins) -> maxphi bits.
template <int MODE>
__global__ __launch_bounds__(256) void k_reduce_partials(State st, const double* __restrict__ partial, int ntiles,
                                                         int nvalid, int s0, const int* __restrict__ active) {
    __shared__ double red[256];
    const int s = s0 + blockIdx.x;
    if (!active[s]) return;
    const double* __restrict__ p = partial + (size_t)s * ntiles;
    double acc = 0.0;
    for (int i = threadIdx.x; i < nvalid; i += 256) {   // nvalid <= ntiles: tiles the producing pass really has
        const double v = p[i];
        if (MODE == 0) acc += v;
        else acc = (v != v) ? v : ((acc != acc) ? acc : (v > acc ? v : acc));
    }
    const double r = MODE == 0 ? block_sum<256>(acc, red) : block_max<256>(acc, red);
    if (threadIdx.x == 0) {
        if (MODE == 0) st.alias_sum[s] = r;
        else st.maxphi[s] = (unsigned long long)__double_as_longlong(r);
    }
}

// ---------------------------------------------------------------------------
// scalar kernels: one thread per stream, arithmetic order of the reference
// ---------------------------------------------------------------------------
// get_timestep (simulation_object.rs:878-934; expanding :939-990)
__global__ void k_timestep(State st, Consts c, int s0, int nb) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int s = s0 + i;
    if (!st.active[s]) return;
    const double pmax_d = __longlong_as_double((long long)st.maxphi[s]);
    st.maxphi_step[s] = pmax_d;
    const real pmax = (real)pmax_d;
    const real two = (real)2.0, twopi = (real)(2.0 * 3.14159265358979323846);
    real kinetic, potential, ttnd;
    if (c.expanding) {
        kinetic = c.cfl * two * c.box / sqrt(c.k2max);
        potential = c.cfl * twopi / ((real)(2.0 * (double)st.a_now[s]) * pmax);
        ttnd = st.tau_next[s];
    } else {
        kinetic = c.cfl * two * c.box / sqrt(c.k2max) / c.hbar;
        potential = c.cfl * twopi * c.hbar / (two * pmax);
        ttnd = (real)(st.dumps[s] + 1) * c.tfinal / (real)c.ndumps - st.time[s];
    }
    real dt = fmin(fmin(kinetic, potential), ttnd);
    st.dt[s] = dt;
    st.dump[s] = (dt == ttnd) ? 1 : 0;
    if (!c.expanding) st.kick[s] = -dt / c.hbar;
}

// kinetic half-drift table exp(i c k_i^2 (2 pi)^2), c = -dt/4 * hbar_
// (simulation_object.rs:504-516; expanding c = -dtau/4, :699-708).  The k-phase
// is separable over the axes, so each pass multiplies by one 1-D factor.
// Evaluated in double for both precisions.
__global__ void k_drift_table(State st, Consts c, const real* __restrict__ ksq, int s0) {
    const int s = s0 + blockIdx.y;
    if (!st.active[s]) return;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= c.N) return;
    real coef;
    {
#pragma clang fp contract(off)
        const real dt = st.dt[s];
        coef = c.expanding ? (-dt / (real)4.0) : (-dt / (real)4.0 * c.hbar);
    }
    const real k2 = ksq[i] * c.fourpi2;
    double sn, cs;
    sincos((double)coef * (double)k2, &sn, &cs);
    st.drift[(size_t)s * c.N + i] = mk<real>((real)cs, (real)sn);
}

// tail of update (simulation_object.rs:589-635; expanding :757-759,:814-844)
__global__ void k_post_step(State st, Consts c, int s0, int nb, const real* new_time, const real* new_tau,
                            const real* snapped_tau) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nb) return;
    const int s = s0 + i;
    if (!st.active[s]) return;
    if (c.expanding) {
        st.time[s] = new_time[s];
        st.tau[s] = new_tau[s];
    } else {
        st.time[s] = st.time[s] + st.dt[s];
    }
    const real p_mass = (real)st.alias_sum[s] * c.dkpow;
    const bool aliased = p_mass > c.alias_thr;
    st.aliased[s] = aliased ? 1 : 0;
    if (!aliased && st.dump[s]) {
        st.dumps[s] += 1;
        st.time[s] = (real)st.dumps[s] * c.tfinal / (real)c.ndumps;
        if (c.expanding) st.tau[s] = snapped_tau[s];
    }
    st.nsteps[s] += 1;
    st.active[s] = (!aliased && st.time[s] < c.tfinal) ? 1 : 0;
}

__global__ void k_publish(State st, Consts c, int ns, int fresh_max) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= ns) return;
    msm_stream_status o;
    o.time = (double)st.time[s];
    o.dt = (double)st.dt[s];
    o.max_phi = fresh_max ? __longlong_as_double((long long)st.maxphi[s]) : st.maxphi_step[s];
    o.alias_mass = (double)((real)st.alias_sum[s] * c.dkpow);
    o.tau = (double)st.tau[s];
    o.n_steps = st.nsteps[s];
    o.current_dumps = st.dumps[s];
    o.dump = st.dump[s];
    o.active = st.active[s];
    o.aliased = st.aliased[s];
    st.status[s] = o;
}

// ---------------------------------------------------------------------------
// initial-condition helpers
// ---------------------------------------------------------------------------
__global__ void k_interleave(cplx* __restrict__ dst, const real* __restrict__ re, const real* __restrict__ im, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = mk<real>(re[i], im[i]);
}
__global__ void k_split(const cplx* __restrict__ src, real* __restrict__ re, real* __restrict__ im, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const cplx v = src[i];
        re[i] = v.x;
        im[i] = v.y;
    }
}

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Wigner / Husimi perturbation (sample_quantum_perturbation, ics.rs:560-646):
// psi <- (psi sqrt(dV) + (N1 + i N2) / div) / sqrt(dV).  Element e draws its
// pair from Philox4x32-10 counter (e_lo, e_hi, 0, 0), key = seed, Box-Muller in
// double.  (ArrayFire's own normal stream is not reproducible from the
// reference tree; statistics, not bits, are what carries over.)
__global__ void k_sample(cplx* __restrict__ psi, size_t n, unsigned long long seed, real sqrt_dv, real div) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        unsigned r[4];
        philox4x32_10((unsigned)(e & 0xffffffffu), (unsigned)(e >> 32), 0u, 0u, (unsigned)(seed & 0xffffffffu),
                      (unsigned)(seed >> 32), r);
        const unsigned long long b1 = (((unsigned long long)r[1] << 32) | r[0]) >> 11;
        const unsigned long long b2 = (((unsigned long long)r[3] << 32) | r[2]) >> 11;
        const double u1 = ((double)b1 + 0.5) * 0x1.0p-53, u2 = ((double)b2 + 0.5) * 0x1.0p-53;
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincos(2.0 * 3.14159265358979323846 * u2, &sn, &cs);
        const cplx z = mk<real>((real)(rad * cs), (real)(rad * sn));
        cplx v = psi[e];
        v = mk<real>(v.x * sqrt_dv, v.y * sqrt_dv);
        v = mk<real>(v.x + z.x / div, v.y + z.y / div);
        psi[e] = mk<real>(v.x / sqrt_dv, v.y / sqrt_dv);
    }
}

// Poisson resampling (sample_quantum_perturbation, ics.rs:495-558): count ~ Poisson(|psi|^2 dV n) per cell,
// psi <- sqrt(count / n) e^{i arg psi} / sqrt(dV).  The reference draws from a thread-local host RNG
// that cannot be reproduced; here element e draws from its own Philox4x32-10 stream, counter
// (e_lo, e_hi, j, 1), key = seed: Knuth's product method below lambda = 10, Hoermann's PTRS
// transformed rejection above.
__device__ __forceinline__ void philox_u2(size_t e, unsigned j, unsigned lane, unsigned long long seed, double* u1, double* u2) {
    unsigned r[4];
    philox4x32_10((unsigned)(e & 0xffffffffu), (unsigned)(e >> 32), j, lane, (unsigned)(seed & 0xffffffffu),
                  (unsigned)(seed >> 32), r);
    const unsigned long long b1 = (((unsigned long long)r[1] << 32) | r[0]) >> 11;
    const unsigned long long b2 = (((unsigned long long)r[3] << 32) | r[2]) >> 11;
    *u1 = ((double)b1 + 0.5) * 0x1.0p-53;
    *u2 = ((double)b2 + 0.5) * 0x1.0p-53;
}
__global__ void k_sample_poisson(cplx* __restrict__ psi, size_t n, unsigned long long seed, double dv, double npart) {
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const cplx v = psi[e];
        const double m2 = (double)v.x * (double)v.x + (double)v.y * (double)v.y;
        const double lam = (m2 * dv) * npart;
        double k = 0.0;
        if (lam < 10.0) {
            const double limit = exp(-lam);
            double p = 1.0;
            for (unsigned j = 0; j < 64; ++j) {
                double u1, u2;
                philox_u2(e, j, 1u, seed, &u1, &u2);
                p *= u1;
                if (!(p > limit)) break;
                k += 1.0;
                p *= u2;
                if (!(p > limit)) break;
                k += 1.0;
            }
        } else {
            const double slam = sqrt(lam), loglam = log(lam);
            const double b = 0.931 + 2.53 * slam;
            const double a = -0.059 + 0.02483 * b;
            const double invalpha = 1.1239 + 1.1328 / (b - 3.4);
            const double vr = 0.9277 - 3.6224 / (b - 2.0);
            for (unsigned j = 0; j < 64; ++j) {
                double u1, V;
                philox_u2(e, j, 1u, seed, &u1, &V);
                const double U = u1 - 0.5, us = 0.5 - fabs(U);
                k = floor((2.0 * a / us + b) * U + lam + 0.43);
                if (us >= 0.07 && V <= vr) break;
                if (k < 0.0 || (us < 0.013 && V > us)) { k = 0.0; continue; }
                if (log(V) + log(invalpha) - log(a / (us * us) + b) <= -lam + k * loglam - lgamma(k + 1.0)) break;
                k = 0.0;
            }
        }
        const double amp = sqrt(k / npart) / sqrt(dv);
        const double mag = sqrt(m2);
        const double ux = mag > 0.0 ? (double)v.x / mag : 1.0, uy = mag > 0.0 ? (double)v.y / mag : 0.0;
        psi[e] = mk<real>((real)(amp * ux), (real)(amp * uy));
    }
}

// sum over streams of |psi_s|^2 (double accumulation)
__global__ void k_density_sum(const cplx* __restrict__ X, size_t n3, int ns, double* __restrict__ out, int accumulate) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x) {
        double acc = accumulate ? out[i] : 0.0;
        for (int s = 0; s < ns; ++s) {
            const cplx v = X[(size_t)s * n3 + i];
            acc += (double)v.x * (double)v.x + (double)v.y * (double)v.y;
        }
        out[i] = acc;
    }
}

// ---------------------------------------------------------------------------
// Pointwise operators of the unfused step: 1-D and 2-D grids (tiny next to the 3-D ones) and 3-D
// grids whose edge is not a power of two up to 512 (chirp-z line transforms below).  Their step
// is the reference's own sequence -- whole transforms (plain axis passes) with one kernel per
// pointwise operator in between -- rather than the fused 3-D pipelines.
// blockIdx.y = local stream; cell i = x + N y + N^2 z.
// ---------------------------------------------------------------------------
struct PwArgs {
    const cplx* src;     // [.][n]
    cplx* dst;
    cplx* W;             // [batch][n] Poisson work grid, one stream per slot
    real* PHI;
    size_t n;
    const int* active;
    int s0, src_local, dst_local, N, dims;
    const cplx* drift;
    const real* ksq;
    const real* kick;
    real scale, mass_factor, pois_c, fourpi2, inv_n, alias_cut;
    double* partial;     // [n_streams][ntiles]
    int ntiles;
};
// cell i -> (x, y, z), x fastest; y and z are 0 on the axes a smaller grid does not have
__device__ __forceinline__ void pw_xyz(const PwArgs& a, size_t i, int& x, int& y, int& z) {
    const size_t r = i / a.N;
    x = (int)(i - r * a.N);
    z = (int)(r / a.N);
    y = (int)(r - (size_t)z * a.N);
}
// spec_grid (utils/fft.rs:123-161): the axes are summed in order, then times (2 pi)^2
__device__ __forceinline__ real pw_k2(const PwArgs& a, int x, int y, int z) {
    real k2 = a.ksq[x];
    if (a.dims >= 2) k2 = k2 + a.ksq[y];
    if (a.dims >= 3) k2 = k2 + a.ksq[z];
    return k2 * a.fourpi2;
}
// MODE 0: dst = src * D (kinetic half-step, simulation_object.rs:504-516)
// MODE 1: dst = src * D * scale, plus this block's aliased-mass partial (:562-574, :1249-1280)
template <int MODE>
__global__ __launch_bounds__(256) void k_pw_drift(PwArgs a) {
    __shared__ double red[256];
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (!a.active[sg]) return;
    const cplx* __restrict__ src = a.src + (size_t)(a.src_local ? ls : sg) * a.n;
    cplx* __restrict__ dst = a.dst + (size_t)(a.dst_local ? ls : sg) * a.n;
    const cplx* __restrict__ d = a.drift + (size_t)sg * a.N;
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        int x, y, z;
        pw_xyz(a, i, x, y, z);
        cplx f = d[x];
        if (a.dims >= 2) f = cmul(f, d[y]);
        if (a.dims >= 3) f = cmul(f, d[z]);
        cplx v = cmul(src[i], f);
        if constexpr (MODE == 1) {
            v = cscale(v, a.scale);
            if (pw_k2(a, x, y, z) > a.alias_cut) acc += (double)(v.x * v.x + v.y * v.y);
        }
        dst[i] = v;
    }
    if constexpr (MODE == 1) {
        const double tot = block_sum<256>(acc, red);
        if (threadIdx.x == 0) a.partial[(size_t)sg * a.ntiles + blockIdx.x] = tot;
    }
}
// W = M |psi * scale|^2 (calculate_density, :1031-1063); src = psi (scale 1) or the unscaled inverse transform
__global__ __launch_bounds__(256) void k_pw_rho(PwArgs a) {
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (!a.active[sg]) return;
    const cplx* __restrict__ src = a.src + (size_t)(a.src_local ? ls : sg) * a.n;
    cplx* __restrict__ w = a.W + (size_t)ls * a.n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        const cplx v = cscale(src[i], a.scale);
        w[i] = mk<real>(a.mass_factor * (v.x * v.x + v.y * v.y), (real)0);
    }
}
// W *= -C / k^2 / N^dims, k = 0 -> 0 (:1076-1102)
__global__ __launch_bounds__(256) void k_pw_green(PwArgs a) {
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (!a.active[sg]) return;
    cplx* __restrict__ w = a.W + (size_t)ls * a.n;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        int x, y, z;
        pw_xyz(a, i, x, y, z);
        const real k2 = pw_k2(a, x, y, z);
        const real g = (k2 == (real)0) ? (real)0 : (a.pois_c * recip_t(k2)) * a.inv_n;
        w[i] = cscale(w[i], g);
    }
}
// phi = Re W: per-block max|phi| (:905), optionally stored (msm_potential)
template <int STOREPHI>
__global__ __launch_bounds__(256) void k_pw_phi(PwArgs a) {
    __shared__ double red[256];
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (!a.active[sg]) return;
    const cplx* __restrict__ w = a.W + (size_t)ls * a.n;
    real* __restrict__ phi = a.PHI + (size_t)ls * a.n;
    double m = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        const real p = w[i].x;
        if constexpr (STOREPHI) phi[i] = p;
        m = nanmax(fabs((double)p), m);
    }
    const double tot = block_max<256>(m, red);
    if (threadIdx.x == 0) a.partial[(size_t)sg * a.ntiles + blockIdx.x] = tot;
}
// psi = (psi * scale) * exp(i kick Re W) (:535-545), in place on dst
__global__ __launch_bounds__(256) void k_pw_kick(PwArgs a) {
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (!a.active[sg]) return;
    cplx* __restrict__ x = a.dst + (size_t)ls * a.n;
    const cplx* __restrict__ w = a.W + (size_t)ls * a.n;
    const real kc = a.kick[sg];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (size_t)gridDim.x * 256) {
        real sn, cs;
        sincos_t(kc * w[i].x, &sn, &cs);
        x[i] = cmul(cscale(x[i], a.scale), mk<real>(cs, sn));
    }
}

// ---------------------------------------------------------------------------
// Line transform of ANY even length N <= 1024 (the reference only asks for an even edge,
// utils/fft.rs:105; ArrayFire transforms any length): Bluestein's chirp-z algorithm,
//   X[k] = c[k] * sum_n (x[n] c[n]) conj(c)[k - n],   c[n] = exp(-i pi n^2 / N),
// the convolution done with power-of-two transforms of length M >= 2N - 1 held in LDS:
// decimation in frequency forward (natural in, bit-reversed out), times the precomputed
// spectrum of conj(c) (stored bit-reversed, 1/M folded in), decimation in time back
// (bit-reversed in, natural out).  A workgroup owns `lpw` neighbouring lines.  The inverse
// transform is conj(F[conj(x)]).  Unnormalised; `scale` is applied on store.
// ---------------------------------------------------------------------------
struct AnyArgs {
    const cplx* src;
    cplx* dst;
    size_t n3;
    const int* active;   // may be null
    int s0, src_local, dst_local;
    int N, M, lpw;
    size_t stride;       // N^axis: distance between consecutive points of a line
    const cplx* chirp;   // [N]    exp(-i pi n^2 / N)
    const cplx* bhat;    // [M]    F_M[conj chirp, wrapped] / M in bit-reversed order
    const cplx* twm;     // [M/2]  exp(-2 pi i k / M)
    real scale;
};
template <int SGN>
__global__ __launch_bounds__(256) void k_line_any(AnyArgs a) {
    extern __shared__ __align__(16) unsigned char any_smem[];
    cplx* buf = reinterpret_cast<cplx*>(any_smem);            // [lpw][M]
    const int ls = blockIdx.y, sg = a.s0 + ls;
    if (a.active != nullptr && !a.active[sg]) return;
    const cplx* __restrict__ src = a.src + (size_t)(a.src_local ? ls : sg) * a.n3;
    cplx* __restrict__ dst = a.dst + (size_t)(a.dst_local ? ls : sg) * a.n3;
    const int N = a.N, M = a.M, lpw = a.lpw, tid = threadIdx.x;
    // line l = lo + stride * hi starts at hi * stride * N + lo; the lpw lines of a workgroup share hi
    const size_t l0 = (size_t)blockIdx.x * lpw;
    const size_t hi = l0 / a.stride, lo = l0 - hi * a.stride;
    const size_t base = hi * a.stride * N + lo;
    const bool xaxis = a.stride == 1;

    for (int e = tid; e < lpw * M; e += 256) {
        int ln, j;
        if (xaxis) { ln = e / M; j = e - ln * M; } else { j = e / lpw; ln = e - j * lpw; }
        cplx v = mk<real>(0, 0);
        if (j < N) {
            v = src[base + (xaxis ? (size_t)ln * N + j : (size_t)j * a.stride + ln)];
            if (SGN > 0) v.y = -v.y;
            v = cmul(v, a.chirp[j]);
        }
        buf[ln * M + j] = v;
    }
    __syncthreads();
    const int halfM = M >> 1, nbf = lpw * halfM;
    for (int half = halfM, tstep = 1; half >= 1; half >>= 1, tstep <<= 1) {
        for (int b = tid; b < nbf; b += 256) {
            const int ln = b / halfM, r = b - ln * halfM;
            const int j = r & (half - 1), g = r / half;
            const int i0 = ln * M + g * 2 * half + j, i1 = i0 + half;
            const cplx u = buf[i0], v = buf[i1];
            buf[i0] = cadd(u, v);
            buf[i1] = cmul(csub(u, v), a.twm[j * tstep]);
        }
        __syncthreads();
    }
    for (int e = tid; e < lpw * M; e += 256) buf[e] = cmul(buf[e], a.bhat[e % M]);
    __syncthreads();
    for (int half = 1, tstep = halfM; half <= halfM; half <<= 1, tstep >>= 1) {
        for (int b = tid; b < nbf; b += 256) {
            const int ln = b / halfM, r = b - ln * halfM;
            const int j = r & (half - 1), g = r / half;
            const int i0 = ln * M + g * 2 * half + j, i1 = i0 + half;
            cplx w = a.twm[j * tstep];
            w.y = -w.y;
            const cplx u = buf[i0], v = cmul(buf[i1], w);
            buf[i0] = cadd(u, v);
            buf[i1] = csub(u, v);
        }
        __syncthreads();
    }
    for (int e = tid; e < lpw * N; e += 256) {
        int ln, j;
        if (xaxis) { ln = e / N; j = e - ln * N; } else { j = e / lpw; ln = e - j * lpw; }
        cplx v = cmul(buf[ln * M + j], a.chirp[j]);
        if (SGN > 0) v.y = -v.y;
        dst[base + (xaxis ? (size_t)ln * N + j : (size_t)j * a.stride + ln)] = cscale(v, a.scale);
    }
}

// In-situ synthesizer sums (synthesizer/src/main.rs:63-93): acc = [psi | |psi|^2 | psik | |psik|^2], each n complex
// doubles; psik arrives unitary and is scaled to the synthesizer's un-normalised transform.
__global__ void k_synth_acc(const cplx* __restrict__ psi, const cplx* __restrict__ psik, double2* __restrict__ acc, size_t n,
                            double kscale) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double pr = (double)psi[i].x, pi = (double)psi[i].y;
        const double kr = (double)psik[i].x * kscale, ki = (double)psik[i].y * kscale;
        double2 a = acc[i]; a.x += pr; a.y += pi; acc[i] = a;
        double2 b = acc[n + i]; b.x += pr * pr + pi * pi; acc[n + i] = b;
        double2 c = acc[2 * n + i]; c.x += kr; c.y += ki; acc[2 * n + i] = c;
        double2 d = acc[3 * n + i]; d.x += kr * kr + ki * ki; acc[3 * n + i] = d;
    }
}
// mean of one field to two real planes
__global__ void k_synth_mean(const double2* __restrict__ f, double* __restrict__ re, double* __restrict__ im, size_t n, double inv) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = f[i];
        re[i] = v.x * inv;
        im[i] = v.y * inv;
    }
}
// per-block partials of sum( <psi2> - |<psi>|^2 ), complex (main.rs:161-175)
__global__ __launch_bounds__(256) void k_synth_qx(const double2* __restrict__ acc, size_t n, double inv, double2* __restrict__ partial) {
    __shared__ double red[256];
    double sr = 0.0, si = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double2 p = acc[i], p2 = acc[n + i];
        const double mr = p.x * inv, mi = p.y * inv;
        sr += p2.x * inv - (mr * mr + mi * mi);
        si += p2.y * inv;
    }
    const double tr = block_sum<256>(sr, red);
    __syncthreads();
    const double ti = block_sum<256>(si, red);
    if (threadIdx.x == 0) { partial[blockIdx.x].x = tr; partial[blockIdx.x].y = ti; }
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
    int synth_release(int slot, std::string& err) override;
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
    void lowdim_poisson(const cplx* psi, bool local, real psi_scale, int s0, int nb, bool store_phi, const int* act);
    void lowdim_step_back(int s0, int nb, const real* nt, const real* ntau, const real* stau);
    PoisArgs pois_args(int s0, int nb, const int* act) const;
    // W passes after the fused/rho x pass; with_x = false leaves F_y,z-inverted W for a fused consumer
    void poisson_tail(int s0, int nb, bool store_phi, bool track_max, bool with_x = true, const int* act = nullptr);
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
    cplx *chirp_ = nullptr, *bhat_ = nullptr, *twm_ = nullptr;
    template <int SGN> void launch_any(int ax, const PsiArgs& a, int nblk_y, int kind);
    bool two_pass_ = false;  // every 3-D transform in two passes (N = 256); P_ then holds S = pass A of psi between steps
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
    int synth_check(int slot, std::string& err) const {
        if (slot < 0 || slot >= (int)synth_.size() || !synth_[slot].sums) { err = "synthesizer slot is not open"; return MSM_ERR_INVALID; }
        return MSM_OK;
    }
    // psi (natural order) and unitary psi_k (reference order) of stream s on the device; valid until the next call
    void materialise(int s, const cplx** psi, const cplx** psik);

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

    // sub-batch: bounded scratch (X + W + PHI = 2 complex grids per stream), whole Poisson pairs
    batch_ = cfg_.batch_streams > 0 ? cfg_.batch_streams : 32;
    if (batch_ < ns_ && (batch_ & 1)) batch_ += 1;
    if (batch_ > ns_) batch_ = ns_;
    if (N_ == tp::N && dims_ == 3 && batch_ > tp::MAX_BATCH) batch_ = tp::MAX_BATCH;

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
    if (const char* e = getenv("MSM_TWO_PASS")) two_pass_ = two_pass_ && atoi(e) != 0;   // 0: the generic three-pass engine
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
    if (anyn_) {
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
    HIP_TRY(hipMemcpy(st_.nsteps, zl.data(), sizeof(long long) * ns_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemcpy(st_.drift, one_c.data(), sizeof(cplx) * (size_t)ns_ * N_, hipMemcpyHostToDevice));
    HIP_TRY(hipMemset(P_, 0, grid_bytes * ns_));
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

// y, z (fused with the Green's multiply), y^-1, x^-1 of the Poisson solve; W already holds F_x[rho]
// (two-pass layout: W holds T and one pass-B kernel replaces the three middle passes)
void Engine::poisson_tail(int s0, int nb, bool store_phi, bool track_max, bool with_x, const int* act) {
    if (!act) act = st_.active;
    const int pairs = (nb + 1) / 2;
    const PoisArgs p = pois_args(s0, nb, act);
    if (two_pass_) {
        {
            timed(PK_POISSON_Z, [&] { k_poisson_b<<<dim3(tp::B_BLOCKS, (unsigned)pairs), dim3(tp::B_THREADS), 0, stream_>>>(p); });
            const dim3 grid(tp::A_TILES, (unsigned)pairs), block(tp::A_THREADS);
            if (with_x) timed(PK_PHI_X, [&] {
                if (store_phi && track_max) k_phi_out2<1, 1><<<grid, block, 0, stream_>>>(p);
                else if (store_phi) k_phi_out2<1, 0><<<grid, block, 0, stream_>>>(p);
                else k_phi_max3<<<dim3(tp::A_TILES), block, 0, stream_>>>(p);
            });
        }
        if (with_x && track_max) k_reduce_partials<1><<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, tp::A_TILES, s0, act);
        if (with_x && store_phi) { phi_s0_ = s0; phi_nb_ = nb; }
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
    if (with_x && track_max) k_reduce_partials<1><<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, tiles_x_, s0, act);
    if (with_x && store_phi) { phi_s0_ = s0; phi_nb_ = nb; }
}

// whole Poisson solve from a psi that sits in memory
void Engine::poisson_from_psi(const cplx* X, bool local, int s0, int nb, bool store_phi, bool track_max, const int* act) {
    PoisArgs p = pois_args(s0, nb, act);
    p.X = X; p.x_local = local ? 1 : 0;
    if (two_pass_) {
                    timed(PK_RHO_X, [&] { k_rho_pass2<<<dim3(tp::A_TILES, (unsigned)((nb + 1) / 2)), dim3(tp::A_THREADS), 0, stream_>>>(p); });
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
    b.scale = c_.unit_scale; b.alias_cut = c_.alias_cut; b.fourpi2 = c_.fourpi2; b.partial = partial_; b.ntiles = ntiles_;
    const dim3 grid(tp::B_BLOCKS, (unsigned)n), block(tp::B_THREADS);
    timed(kind, [&] {
        if (alias) k_psi_rt<1><<<grid, block, 0, stream_>>>(b);
        else k_psi_rt<0><<<grid, block, 0, stream_>>>(b);
    });
}

void Engine::psi_b_fwd(const cplx* src, cplx* dst, int s0, int n) {
    PsiBArgs b{};
    b.src = src; b.dst = dst; b.n3 = n3_; b.active = nullptr; b.s0 = s0; b.tw = tw_; b.scale = c_.unit_scale;
    timed(PK_PLAIN, [&] { k_psi_b_fwd<<<dim3(tp::B_BLOCKS, (unsigned)n), dim3(tp::B_THREADS), 0, stream_>>>(b); });
}

void Engine::psi_a(int sgn, const cplx* src, bool src_local, cplx* dst, bool dst_local, int s0, int n, const int* act) {
    PsiArgs a{};
    a.src = src; a.dst = dst; a.n3 = n3_; a.active = act; a.s0 = s0; a.src_local = src_local; a.dst_local = dst_local;
    a.tw = tw_; a.scale = c_.unit_scale;
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
void Engine::lowdim_poisson(const cplx* psi, bool local, real psi_scale, int s0, int nb, bool store_phi, const int* act) {
    PwArgs a = pw_args(s0, act);
    a.src = psi; a.src_local = local ? 1 : 0; a.scale = psi_scale;
    const dim3 grid(pw_blocks_, (unsigned)nb), block(256);
    k_pw_rho<<<grid, block, 0, stream_>>>(a);
    fft3_plain(W_, true, W_, true, s0, nb, -1, act, false);
    k_pw_green<<<grid, block, 0, stream_>>>(a);
    fft3_plain(W_, true, W_, true, s0, nb, +1, act, false);
    if (store_phi) k_pw_phi<1><<<grid, block, 0, stream_>>>(a);
    else k_pw_phi<0><<<grid, block, 0, stream_>>>(a);
    k_reduce_partials<1><<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, pw_blocks_, s0, act);
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
    lowdim_poisson(X_, true, (real)1, s0, nb, false, st_.active);
    a.dst = X_; a.scale = (real)1;
    k_pw_kick<<<grid, block, 0, stream_>>>(a);
    // psi_k = D F[psi], aliased mass
    fft3_plain(X_, true, X_, true, s0, nb, -1, st_.active);
    a.src = X_; a.src_local = 1; a.dst = P_; a.dst_local = 0; a.scale = (real)1;
    k_pw_drift<1><<<grid, block, 0, stream_>>>(a);
    k_reduce_partials<0><<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, pw_blocks_, s0, st_.active);
    k_post_step<<<(nb + 63) / 64, 64, 0, stream_>>>(st_, c_, s0, nb, nt, ntau, stau);
    // potential of psi(t + dt) for the next timestep (:497)
    fft3_plain(P_, false, X_, true, s0, nb, +1, st_.active);
    lowdim_poisson(X_, true, (real)1, s0, nb, false, st_.active);
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

// timestep (simulation_object.rs:500); max|phi| of psi(t) was left by the previous step's tail
void Engine::step_front(int s0, int nb) {
    k_timestep<<<(nb + 63) / 64, 64, 0, stream_>>>(st_, c_, s0, nb);
}

// drift/2, F^-1, potential, kick, F, drift/2, F^-1, bookkeeping, potential for the next dt (:504-635, :497)
void Engine::step_back(int s0, int nb, const real* nt, const real* ntau, const real* stau) {
    k_drift_table<<<dim3((N_ + 63) / 64, nb), 64, 0, stream_>>>(st_, c_, ksq_, s0);
    PsiArgs a{};
    a.n3 = n3_; a.active = st_.active; a.s0 = s0; a.pair_nb = 0; a.tw = tw_; a.drift = st_.drift;
    a.phi = PHI_; a.kick = st_.kick; a.scale = (real)1;
    PoisArgs p = pois_args(s0, nb, st_.active);

    if (lowdim_) { lowdim_step_back(s0, nb, nt, ntau, stau); return; }
    if (two_pass_) {
        {
            const dim3 gridA(tp::A_TILES), gridAp(tp::A_TILES, (unsigned)((nb + 1) / 2)), blockA(tp::A_THREADS);
            // The whole step runs in place on the batch's states S (contiguous in P_): every kernel reads a tile or a
            // run completely before it stores it, and tiles and runs are disjoint between workgroups.
            cplx* S = P_ + (size_t)s0 * n3_;
            p.X = S; p.Xout = S;
            // psi = F^-1[ D psi_k ]: round trip of S through pass B with the whole drift factor in between, then
            // pass A^-1 fused with rho and the Poisson pass A
            psi_rt(S, S, false, s0, nb, st_.active, PK_INV_Z);
            timed(PK_INVX_RHO, [&] { k_invx_rho3<0><<<gridA, blockA, 0, stream_>>>(p); });
            poisson_tail(s0, nb, false, false, false);
            // kick between the two A passes; the second round trip applies the second drift, sums the aliased mass
            // and leaves the new state
            timed(PK_FWD_X, [&] { k_kick3<<<gridAp, blockA, 0, stream_>>>(p, st_.kick); });
            psi_rt(S, S, true, s0, nb, st_.active, PK_FWDZ_INVZ);
            k_reduce_partials<0><<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, tp::B_BLOCKS, s0, st_.active);
            k_post_step<<<(nb + 63) / 64, 64, 0, stream_>>>(st_, c_, s0, nb, nt, ntau, stau);
            p.dens = step_dens_;               // potential of psi(t + dt) straight from the state
            timed(PK_INVX_RHO, [&] {
                if (step_dens_) k_invx_rho3<1><<<gridA, blockA, 0, stream_>>>(p);
                else k_invx_rho3<0><<<gridA, blockA, 0, stream_>>>(p);
            });
            poisson_tail(s0, nb, false, true);
        }
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
    k_reduce_partials<0><<<nb, 256, 0, stream_>>>(st_, partial_, ntiles_, tiles_z_, s0, st_.active);
    k_post_step<<<(nb + 63) / 64, 64, 0, stream_>>>(st_, c_, s0, nb, nt, ntau, stau);
    // streams that are still running: finish psi(t + dt) only as far as the next step needs it,
    // i.e. its density (and the ensemble density if asked for); psi itself is not written back
    launch_psi<+1, 0, 0>(1, a, nb, PK_INV_Y);
    p.dens = step_dens_;
    dispatch_n(N_, [&](auto nc) {
        constexpr int N = decltype(nc)::value;
        using G0 = Geom<N, 0>;
        const dim3 grid((unsigned)(n3_ / N / G0::LPW)), block(G0::THREADS);
        timed(PK_INVX_RHO, [&] {
            if (step_dens_) k_invx_rho<N, 0, 0, 1><<<grid, block, 0, stream_>>>(p);
            else k_invx_rho<N, 0, 0, 0><<<grid, block, 0, stream_>>>(p);
        });
    });
    poisson_tail(s0, nb, false, true);
}

int Engine::step(std::string& err) {
    if (cfg_.expanding) { err = "expanding contexts step with msm_step_begin / msm_step_end"; return MSM_ERR_STATE; }
    if (in_step_) { err = "msm_step inside an open msm_step_begin"; return MSM_ERR_STATE; }
    phi_s0_ = -1;   // the batch potential buffer is not maintained by steps
    enter_fourier();
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
    if (graph_ok_ && !prof_on_ && n3_ * (size_t)ns_ <= GRAPH_MAX_CELLS) {
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
    return MSM_OK;
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
    if (step_dens_) HIP_TRY(hipMemsetAsync(step_dens_, 0, sizeof(double) * n3_, stream_));
    for_batches([&](int s0, int nb) { step_back(s0, nb, d_in_[0], d_in_[1], d_in_[2]); });
    publish();
    in_step_ = false;
    HIP_TRY(hipGetLastError());
    return MSM_OK;
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
            timed(PK_INVX_RHO, [&] { k_invx_rho3<0><<<dim3(tp::A_TILES), dim3(tp::A_THREADS), 0, stream_>>>(p); });
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
    (void)err;
    step_dens_ = (double*)dev_out;
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
