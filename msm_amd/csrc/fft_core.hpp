// Batched power-of-two complex FFT building blocks for gfx950 (wave64).
//
// One "line" of N points is owned by TPL = N/E threads; every thread keeps E
// complex values in registers.  The transform is a Stockham autosort
// factorisation N = R0*R1[*R2] (radices <= 16): each stage does E/R
// register-resident radix-R butterflies per thread and, except after the last
// stage, exchanges the line through LDS.  With this indexing a thread always
// reads positions  j0 + TPL*q  (q = 0..E-1)  -- from global memory before the
// first stage, from LDS before the later ones -- and after the last stage it
// holds the natural-order outputs at those same positions, so the global
// load and store patterns are identical and coalesce in runs of TPL elements
// (contiguous axis) or LPW elements (strided axes, LPW neighbouring lines).
//
// Replaces the ArrayFire fft*/ifft* calls behind the reference's wrappers
// (simulator/src/utils/fft.rs:6-98); sign convention forward = e^{-2 pi i kn/N}.
#pragma once
#include <hip/hip_runtime.h>

namespace msm {

template <typename T> struct cx_of;
template <> struct cx_of<double> { using type = double2; };
template <> struct cx_of<float>  { using type = float2; };
template <typename T> using cx = typename cx_of<T>::type;

template <typename T> __device__ __forceinline__ cx<T> mk(T x, T y) { cx<T> r; r.x = x; r.y = y; return r; }
template <typename C> __device__ __forceinline__ C cadd(C a, C b) { C r; r.x = a.x + b.x; r.y = a.y + b.y; return r; }
template <typename C> __device__ __forceinline__ C csub(C a, C b) { C r; r.x = a.x - b.x; r.y = a.y - b.y; return r; }
template <typename C> __device__ __forceinline__ C cmul(C a, C b) { C r; r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; return r; }
// a * conj(b)
template <typename C> __device__ __forceinline__ C cmulc(C a, C b) { C r; r.x = a.x * b.x + a.y * b.y; r.y = a.y * b.x - a.x * b.y; return r; }
template <typename C, typename T> __device__ __forceinline__ C cscale(C a, T s) { C r; r.x = a.x * s; r.y = a.y * s; return r; }

// Single precision: the same operations on both components at once.  gfx950 executes packed fp32 add / mul / fma on a
// register pair at the rate of the scalar forms, so a complex add is one instruction instead of two and a complex
// multiply two instead of four (the component swap and the sign ride along as operand modifiers).
typedef float f32pair __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32pair pk2(float2 a) { f32pair v; v.x = a.x; v.y = a.y; return v; }
__device__ __forceinline__ float2 unpk2(f32pair v) { float2 r; r.x = v.x; r.y = v.y; return r; }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return unpk2(pk2(a) + pk2(b)); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return unpk2(pk2(a) - pk2(b)); }
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    const f32pair va = pk2(a), sw = {a.y, a.x}, bx = {b.x, b.x}, by = {-b.y, b.y};
    return unpk2(va * bx + sw * by);
}
__device__ __forceinline__ float2 cmulc(float2 a, float2 b) {
    const f32pair va = pk2(a), sw = {a.y, a.x}, bx = {b.x, b.x}, by = {b.y, -b.y};
    return unpk2(va * bx + sw * by);
}
__device__ __forceinline__ float2 cscale(float2 a, float s) { const f32pair sv = {s, s}; return unpk2(pk2(a) * sv); }
// p = e + i o = (e.x - o.y, e.y + o.x) and m = e - i o = (e.x + o.y, e.y - o.x): one packed add each, the component
// swap of o and the sign riding along as operand modifiers (written as `e + {-o.y, o.x}` the compiler builds the rotated
// pair first, one v_xor and one v_mov per butterfly -- a tenth of the instructions of a single-precision transform
// kernel).  The wait state in front: between a packed instruction and a dependent one the compiler inserts one (s_nop 0)
// wherever it sees both; it does not look inside an asm statement for the operands' producer, so the statement brings
// its own (behind the statement the compiler does insert it).
__device__ __forceinline__ void cbfly_i(float2 e, float2 o, float2& p, float2& m) {
    f32pair rp, rm;
    asm("s_nop 0\n\t"
        "v_pk_add_f32 %0, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]\n\t"
        "v_pk_add_f32 %1, %2, %3 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]"
        : "=&v"(rp), "=&v"(rm) : "v"(pk2(e)), "v"(pk2(o)));
    p = unpk2(rp); m = unpk2(rm);
}
template <typename C> __device__ __forceinline__ void cbfly_i(C e, C o, C& p, C& m) {
    p.x = e.x - o.y; p.y = e.y + o.x;
    m.x = e.x + o.y; m.y = e.y - o.x;
}

// cos/sin(2 pi m / 16), m = 0..7
__device__ __forceinline__ constexpr double cos16(int m) {
    constexpr double c1 = 0.92387953251128675613, c2 = 0.70710678118654752440, c3 = 0.38268343236508977173;
    return m == 0 ? 1.0 : m == 1 ? c1 : m == 2 ? c2 : m == 3 ? c3 : m == 4 ? 0.0 : m == 5 ? -c3 : m == 6 ? -c2 : -c1;
}
__device__ __forceinline__ constexpr double sin16(int m) {
    constexpr double c1 = 0.92387953251128675613, c2 = 0.70710678118654752440, c3 = 0.38268343236508977173;
    return m == 0 ? 0.0 : m == 1 ? c3 : m == 2 ? c2 : m == 3 ? c1 : m == 4 ? 1.0 : m == 5 ? c1 : m == 6 ? c2 : c3;
}

// In-register DFT of R points (R in {1,2,4,8,16}), natural order in and out.
// SGN = -1 forward, +1 inverse.  Radix-2 decimation in time, fully unrolled.
template <int R, int SGN, typename T>
__device__ __forceinline__ void dft(cx<T> (&a)[R]) {
    if constexpr (R == 2) {
        cx<T> t = a[0];
        a[0] = cadd(t, a[1]);
        a[1] = csub(t, a[1]);
    } else if constexpr (R >= 4) {
        constexpr int H = R / 2;
        cx<T> e[H], o[H];
#pragma unroll
        for (int i = 0; i < H; ++i) { e[i] = a[2 * i]; o[i] = a[2 * i + 1]; }
        dft<H, SGN, T>(e);
        dft<H, SGN, T>(o);
#pragma unroll
        for (int k = 0; k < H; ++k) {
            cx<T> t;
            constexpr int STEP = 16 / R;
            const int m = k * STEP;  // twiddle W_R^k = exp(SGN * 2 pi i m / 16)
            if (m == 4) {            // o * (SGN * i)
                if constexpr (SGN > 0) cbfly_i(e[k], o[k], a[k], a[k + H]);
                else cbfly_i(e[k], o[k], a[k + H], a[k]);
                continue;
            }
            if (k == 0) {
                t = o[k];
            } else {
                t = cmul(o[k], mk<T>((T)cos16(m), (T)(SGN * sin16(m))));
            }
            a[k] = cadd(e[k], t);
            a[k + H] = csub(e[k], t);
        }
    }
}

// Factorisation table.  E = values per thread, S = stages.
template <int N> struct Plan;
template <> struct Plan<2>    { static constexpr int E = 2,  S = 1, R0 = 2,  R1 = 1,  R2 = 1; };
template <> struct Plan<4>    { static constexpr int E = 4,  S = 1, R0 = 4,  R1 = 1,  R2 = 1; };
template <> struct Plan<8>    { static constexpr int E = 8,  S = 1, R0 = 8,  R1 = 1,  R2 = 1; };
template <> struct Plan<16>   { static constexpr int E = 16, S = 1, R0 = 16, R1 = 1,  R2 = 1; };
template <> struct Plan<32>   { static constexpr int E = 16, S = 2, R0 = 16, R1 = 2,  R2 = 1; };
template <> struct Plan<64>   { static constexpr int E = 16, S = 2, R0 = 16, R1 = 4,  R2 = 1; };
template <> struct Plan<128>  { static constexpr int E = 16, S = 2, R0 = 16, R1 = 8,  R2 = 1; };
template <> struct Plan<256>  { static constexpr int E = 16, S = 2, R0 = 16, R1 = 16, R2 = 1; };
// LDS twiddle-table addressing.  Entry for butterfly phase index k (0..NS-1) and leg t is
// exp(-2 pi i k t TWS / N).  The natural table tw[k t TWS] makes the lanes of one read (fixed t,
// k = lane) stride by t elements -- multi-way bank conflicts for t = 4, 8, 12.  For N = 256 only the
// second stage has twiddles (k < 16, t < 16), so its table is stored leg-major, tw[16 t + k], and
// one read touches 16 consecutive entries.
template <int N> struct TwLayout {
    static constexpr bool LEG_MAJOR = (N == 256);
    static __device__ __forceinline__ int index(int k, int t, int tws) { return LEG_MAJOR ? (t * 16 + k) : (k * t * tws); }
    // table slot i holds the natural entry source(i)
    static __device__ __forceinline__ int source(int i) { return LEG_MAJOR ? (((i & 15) * (i >> 4)) & (N - 1)) : i; }
};
template <> struct Plan<512>  { static constexpr int E = 16, S = 3, R0 = 16, R1 = 16, R2 = 2; };

// Lines per workgroup: MSM_LPW_X for passes along the contiguous axis, MSM_LPW_S for the
// strided axes (there LPW neighbouring lines make each global row segment LPW*sizeof(complex) bytes).
#ifndef MSM_LPW_X
#define MSM_LPW_X 16
#endif
#ifndef MSM_LPW_S
#define MSM_LPW_S 16
#endif
template <int N, int AX> struct Geom {
    using P = Plan<N>;
    static constexpr int E = P::E;
    static constexpr int TPL = N / E;                  // threads per line
    static constexpr int LPW0 = AX == 0 ? MSM_LPW_X : MSM_LPW_S;
    static constexpr int LPW = N < LPW0 ? N : LPW0;    // lines per workgroup
    static constexpr int THREADS = TPL * LPW;
    // Padded line length in LDS (elements of 16 B in double).  On the strided axes the 16 lanes of a
    // ds_read_b128 group are 16 neighbouring lines at one position: an odd stride spreads them over
    // the 16 slots of the 256-byte bank row.  Along x a group is one line's lanes 0-3,12-15 and the
    // next line's lanes 4-11 (or the complement): a stride that is a multiple of 16 keeps them apart.
    static constexpr int LINE_LDS = AX == 0 ? ((N + (N >> 4) + 15) & ~15) : (N + (N >> 4) + 1);
    static constexpr bool USES_LDS = P::S > 1;
    // LDS elements: LPW padded lines + the N-entry twiddle table.
    static constexpr int LDS_ELEMS = USES_LDS ? (LPW * LINE_LDS + N) : 1;
};

__device__ __forceinline__ int lds_pad(int p) { return p + (p >> 4); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains every outstanding
// global load and store of the wave (s_waitcnt vmcnt(0)); the exchanges here communicate through LDS alone.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
// Exchange synchronisation.  When all threads of a line sit in one wavefront (LOCAL) the LDS
// operations of the exchange are ordered by the wave's own instruction stream: no barrier at all.
template <bool LOCAL> __device__ __forceinline__ void exch_sync() {
    if constexpr (LOCAL) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else lds_barrier();
}

// Runs all stages of one length-N transform on the E register values of this
// thread.  `line` points at this line's padded LDS region, `tw` at the
// LDS-resident table tw[p] = exp(-2 pi i p / N).  Unless LOCAL (a line's threads share a wavefront)
// it contains workgroup barriers and every thread of the workgroup must call it; `tw` must be
// visible to the caller's threads beforehand.
template <typename T, int N, int SGN, bool LOCAL = false>
struct LineFft {
    using P = Plan<N>;
    static constexpr int E = P::E, TPL = N / E;

    template <int SIDX, int NS>
    static __device__ __forceinline__ void stage(cx<T> (&v)[E], int j0, cx<T>* line, const cx<T>* tw) {
        constexpr int R = SIDX == 0 ? P::R0 : (SIDX == 1 ? P::R1 : P::R2);
        constexpr int BPT = E / R;
        constexpr bool LAST = (SIDX == P::S - 1);
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            const int b = j0 + TPL * u;
            const int k = b & (NS - 1);
            cx<T> a[R];
#pragma unroll
            for (int t = 0; t < R; ++t) a[t] = v[u + t * BPT];
            if constexpr (NS > 1) {
                constexpr int TWS = N / (NS * R);
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    const cx<T> w = tw[TwLayout<N>::index(k, t, TWS)];
                    a[t] = (SGN < 0) ? cmul(a[t], w) : cmulc(a[t], w);
                }
            }
            dft<R, SGN, T>(a);
            if constexpr (LAST) {
#pragma unroll
                for (int t = 0; t < R; ++t) v[u + t * BPT] = a[t];
            } else {
                const int o = (b - k) * R + k;
#pragma unroll
                for (int t = 0; t < R; ++t) line[lds_pad(o + t * NS)] = a[t];
            }
        }
        if constexpr (!LAST) {
            exch_sync<LOCAL>();
#pragma unroll
            for (int q = 0; q < E; ++q) v[q] = line[lds_pad(j0 + TPL * q)];
            if constexpr (SIDX + 2 < P::S) exch_sync<LOCAL>();
            stage<SIDX + 1, NS * R>(v, j0, line, tw);
        }
    }

    static __device__ __forceinline__ void run(cx<T> (&v)[E], int j0, cx<T>* line, const cx<T>* tw) {
        stage<0, 1>(v, j0, line, tw);
    }
};

// The same transform for workgroups whose LDS budget holds only one REAL plane of the tile:
// every exchange moves the real parts and then the imaginary parts through `line`
// (LINE padded reals per line).  On return other threads may still be reading `line`.
template <typename T, int N, int SGN, bool LOCAL = false>
struct LineFftSplit {
    using P = Plan<N>;
    static constexpr int E = P::E, TPL = N / E;

    template <int SIDX, int NS>
    static __device__ __forceinline__ void stage(cx<T> (&v)[E], int j0, T* line, const cx<T>* tw) {
        constexpr int R = SIDX == 0 ? P::R0 : (SIDX == 1 ? P::R1 : P::R2);
        constexpr int BPT = E / R;
        constexpr bool LAST = (SIDX == P::S - 1);
        int o[BPT];
#pragma unroll
        for (int u = 0; u < BPT; ++u) {
            const int b = j0 + TPL * u;
            const int k = b & (NS - 1);
            cx<T> a[R];
#pragma unroll
            for (int t = 0; t < R; ++t) a[t] = v[u + t * BPT];
            if constexpr (NS > 1) {
                constexpr int TWS = N / (NS * R);
#pragma unroll
                for (int t = 1; t < R; ++t) {
                    const cx<T> w = tw[TwLayout<N>::index(k, t, TWS)];
                    a[t] = (SGN < 0) ? cmul(a[t], w) : cmulc(a[t], w);
                }
            }
            dft<R, SGN, T>(a);
#pragma unroll
            for (int t = 0; t < R; ++t) v[u + t * BPT] = a[t];
            o[u] = (b - k) * R + k;
        }
        if constexpr (!LAST) {
#pragma unroll
            for (int u = 0; u < BPT; ++u)
#pragma unroll
                for (int t = 0; t < R; ++t) line[lds_pad(o[u] + t * NS)] = v[u + t * BPT].x;
            exch_sync<LOCAL>();
#pragma unroll
            for (int q = 0; q < E; ++q) v[q].x = line[lds_pad(j0 + TPL * q)];
            exch_sync<LOCAL>();
#pragma unroll
            for (int u = 0; u < BPT; ++u)
#pragma unroll
                for (int t = 0; t < R; ++t) line[lds_pad(o[u] + t * NS)] = v[u + t * BPT].y;
            exch_sync<LOCAL>();
#pragma unroll
            for (int q = 0; q < E; ++q) v[q].y = line[lds_pad(j0 + TPL * q)];
            if constexpr (SIDX + 2 < P::S) exch_sync<LOCAL>();
            stage<SIDX + 1, NS * R>(v, j0, line, tw);
        }
    }

    static __device__ __forceinline__ void run(cx<T> (&v)[E], int j0, T* line, const cx<T>* tw) {
        stage<0, 1>(v, j0, line, tw);
    }
};

}  // namespace msm
