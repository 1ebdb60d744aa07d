// HBM bandwidth probe for the access patterns of the FFT pass kernels (measurement tool).
// Each variant reads 16 complex128 per thread in the pass kernels' pattern and writes them back
// to a second buffer; no arithmetic.  Build: hipcc -O3 --offload-arch=gfx950 tools/bw_probe.hip -o build/bw_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int N = 256;

__global__ void k_copy4(const double4* __restrict__ a, double4* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}
__global__ void k_copy2(const double2* __restrict__ a, double2* __restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

// AX 0: lines along x; 1: along y; 2: along z.  One tile = 16 lines x 256 points, 256 threads.
template <int AX, int LDSB>
__global__ __launch_bounds__(256) void k_tile(const double2* __restrict__ a, double2* __restrict__ b, int ntiles, long total) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && ntiles < 0) pad[0] = 1;   // keep the allocation
    const int tid = threadIdx.x;
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const int tile = (int)(w % ntiles);
        const size_t off = (size_t)(w / ntiles) * N * N * N;
        int ln, j0; size_t base, stride;
        if (AX == 0) { j0 = tid % 16; ln = tid / 16; base = (size_t)(tile * 16 + ln) * N; stride = 1; }
        else if (AX == 1) { ln = tid % 16; j0 = tid / 16; base = (size_t)(tile / 16) * N * N + (tile % 16) * 16 + ln; stride = N; }
        else { ln = tid % 16; j0 = tid / 16; base = (size_t)tile * 16 + ln; stride = (size_t)N * N; }
        double2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = a[off + base + (size_t)(j0 + 16 * q) * stride];
#pragma unroll
        for (int q = 0; q < 16; ++q) b[off + base + (size_t)(j0 + 16 * q) * stride] = v[q];
    }
}

// ---- access patterns of a two-pass 3-D transform with a blocked intermediate layout ------------
// T index = (((((z_hi*16 + tp)*16 + b)*64 + kx_hi)*4 + z_lo)*4 + kx_lo)   (z_hi 64, tp 16, b 16, kx_hi 64)
// B tile: fixed (tp, kx_hi): 64 z_hi x 16 b blocks of 16 contiguous elements (256 B) = 16384 elements,
//         one workgroup of 512 threads, 32 elements per thread, copied in place.
template <int LDSB>
__global__ __launch_bounds__(512) void k_btile(double2* __restrict__ a, long total) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && total < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const size_t off = (size_t)(w / 1024) * N * N * N;
        const int tile = (int)(w % 1024), tp = tile / 64, kxh = tile % 64;
        // thread -> (element in block e = tid % 16, block group g = tid / 16 (32 groups)); 32 blocks per thread
        const int e = tid & 15, g = tid >> 4;
        double2 v[32];
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int blk = g + 32 * q;              // 0..1023 = z_hi * 16 + b
            const int zh = blk >> 4, b = blk & 15;
            const size_t idx = ((((size_t)(zh * 16 + tp) * 16 + b) * 64 + kxh) * 16) + e;
            v[q] = a[off + idx];
        }
#pragma unroll
        for (int q = 0; q < 32; ++q) {
            const int blk = g + 32 * q;
            const int zh = blk >> 4, b = blk & 15;
            const size_t idx = ((((size_t)(zh * 16 + tp) * 16 + b) * 64 + kxh) * 16) + e;
            v[q].x += 1.0; a[off + idx] = v[q];
        }
    }
}
// A tile on the T side: fixed (b, z_hi) and ZL consecutive z_lo: over tp (16), kx_hi (64): segments of
// 4*ZL elements; natural side: 16 rows of 4 KiB.  Copies T -> natural buffer.
template <int ZL, int THREADS, int LDSB>
__global__ __launch_bounds__(THREADS) void k_atile(const double2* __restrict__ t, double2* __restrict__ x, long total) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && total < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    constexpr int PER = 4096 * ZL / THREADS;         // elements per thread
    constexpr int SUB = 4 / ZL;                      // sub-tiles per (b, z_hi)
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const size_t off = (size_t)(w / (1024 * SUB)) * N * N * N;
        const int r = (int)(w % (1024 * SUB)), zl0 = (r % SUB) * ZL, tile = r / SUB, zh = tile / 16, b = tile % 16;
        double2 v[PER];
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = tid + THREADS * q;          // 0 .. 4096*ZL-1 : (tp, kx_hi, zl, kx_lo) with (zl,kx_lo) fastest
            const int in = i % (4 * ZL), kxh = (i / (4 * ZL)) % 64, tp = i / (4 * ZL * 64);
            const size_t idx = ((((size_t)(zh * 16 + tp) * 16 + b) * 64 + kxh) * 16) + zl0 * 4 + in;
            v[q] = t[off + idx];
        }
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int i = tid + THREADS * q;          // natural side: (zl, y row tp, x) with x fastest
            const int xx = i % 256, row = (i / 256) % 16, zl = i / 4096;
            const size_t idx = ((size_t)(zh * 4 + zl0 + zl) * 256 + (b + 16 * row)) * 256 + xx;
            x[off + idx] = v[q];
        }
    }
}

// A tile with 4-element (64-B) segments on the T side, one workgroup looping over the four z_lo sub-tiles so
// that the second half of every 128-B line is touched by the same workgroup shortly after the first.
template <int DIR, int LDSB>
__global__ __launch_bounds__(256) void k_atile_loop(double2* __restrict__ t, double2* __restrict__ x, long total) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && total < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const size_t off = (size_t)(w / 1024) * N * N * N;
        const int tile = (int)(w % 1024), zh = tile / 16, b = tile % 16;
        for (int zl = 0; zl < 4; ++zl) {
            double2 v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = tid + 256 * q;
                const int in = i % 4, kxh = (i / 4) % 64, tp = i / 256;
                const size_t it = ((((size_t)(zh * 16 + tp) * 16 + b) * 64 + kxh) * 16) + zl * 4 + in;
                const size_t ix = ((size_t)(zh * 4 + zl) * 256 + (b + 16 * tp)) * 256 + (i % 256);
                v[q] = DIR == 0 ? t[off + it] : x[off + ix];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int i = tid + 256 * q;
                const int in = i % 4, kxh = (i / 4) % 64, tp = i / 256;
                const size_t it = ((((size_t)(zh * 16 + tp) * 16 + b) * 64 + kxh) * 16) + zl * 4 + in;
                const size_t ix = ((size_t)(zh * 4 + zl) * 256 + (b + 16 * tp)) * 256 + (i % 256);
                if (DIR == 0) x[off + ix] = v[q]; else t[off + it] = v[q];
            }
        }
    }
}
template <int LDSB>
__global__ __launch_bounds__(1024) void k_btile1024(double2* __restrict__ a, long total) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && total < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    for (long w = blockIdx.x; w < total; w += gridDim.x) {
        const size_t off = (size_t)(w / 1024) * N * N * N;
        const int tile = (int)(w % 1024), tp = tile / 64, kxh = tile % 64;
        const int e = tid & 15, g = tid >> 4;
        double2 v[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int blk = g + 64 * q;
            const int zh = blk >> 4, b = blk & 15;
            v[q] = a[off + ((((size_t)(zh * 16 + tp) * 16 + b) * 64 + kxh) * 16) + e];
        }
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int blk = g + 64 * q;
            const int zh = blk >> 4, b = blk & 15;
            v[q].x += 1.0;
            a[off + ((((size_t)(zh * 16 + tp) * 16 + b) * 64 + kxh) * 16) + e] = v[q];
        }
    }
}

// ---- layouts for a pair-packed work array: inner block of 8 elements (128 B) -----------------
// B side: a tile pair is one contiguous run of 16384 elements; a 512-thread workgroup owns every other
// 64-B chunk of it (HALF = 0/1).  PART = distance in blockIdx between the two halves of a run.
template <int PART, int LDSB>
__global__ __launch_bounds__(512) void k_bhalf(double2* __restrict__ a, long runs) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && runs < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    // blockIdx -> (run, half): groups of 2*PART blocks hold PART runs
    const long g = blockIdx.x / (2 * PART);
    const int r = blockIdx.x % (2 * PART), half = r / PART;
    const long run = g * PART + r % PART;
    if (run >= runs) return;
    double2* base = a + (size_t)run * 16384;
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 512 * q;               // 0..8191 within the half
        const int chunk = i >> 2, in = i & 3;      // 64-B chunk index within the half
        v[q] = base[(size_t)(2 * chunk + half) * 4 + in];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 512 * q;
        const int chunk = i >> 2, in = i & 3;
        v[q].x += 1.0;
        base[(size_t)(2 * chunk + half) * 4 + in] = v[q];
    }
}
// B half-runs when the 8-element block is [zl(2)][kx(4)] and a half owns kx bit 1: two 32-B pieces per block.
template <int PART, int LDSB>
__global__ __launch_bounds__(512) void k_bhalf32(double2* __restrict__ a, long runs) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && runs < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    const long g = blockIdx.x / (2 * PART);
    const int r = blockIdx.x % (2 * PART), half = r / PART;
    const long run = g * PART + r % PART;
    if (run >= runs) return;
    double2* base = a + (size_t)run * 16384;
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 512 * q;               // 0..8191 within the half
        const int blk = i >> 2, zl = (i >> 1) & 1, lo = i & 1;
        v[q] = base[(size_t)blk * 8 + zl * 4 + half * 2 + lo];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 512 * q;
        const int blk = i >> 2, zl = (i >> 1) & 1, lo = i & 1;
        v[q].x += 1.0;
        base[(size_t)blk * 8 + zl * 4 + half * 2 + lo] = v[q];
    }
}
// A side with 4096-element tiles (one zl, 256 threads): 64-B chunks [kx(4)] of the block, partner (other zl) PART blocks away.
template <int DIR, int PART, int LDSB>
__global__ __launch_bounds__(256) void k_aseg64(double2* __restrict__ t, double2* __restrict__ x, long tiles, long st, long sk,
                                                long sz, long sb) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && tiles < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    const long g = blockIdx.x / (2 * PART);
    const int r = blockIdx.x % (2 * PART), zl = r / PART;
    const long w = g * PART + r % PART;           // (stream, zh, b)
    const size_t off = (size_t)(w / 2048) * N * N * N;
    const int tile = (int)(w % 2048), zh = tile / 16, b = tile % 16;
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 256 * q;               // (tp, kxh, in4)
        const int in = i & 3, kxh = (i >> 2) & 63, tp = i >> 8;
        const size_t it = (size_t)tp * st + (size_t)kxh * sk + (size_t)zh * sz + (size_t)b * sb + zl * 4 + in;
        const int xx = i & 255, row = i >> 8;
        const size_t ix = ((size_t)(zh * 2 + zl) * 256 + (b + 16 * row)) * 256 + xx;
        v[q] = DIR == 0 ? t[off + it] : x[off + ix];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 256 * q;
        const int in = i & 3, kxh = (i >> 2) & 63, tp = i >> 8;
        const size_t it = (size_t)tp * st + (size_t)kxh * sk + (size_t)zh * sz + (size_t)b * sb + zl * 4 + in;
        const int xx = i & 255, row = i >> 8;
        const size_t ix = ((size_t)(zh * 2 + zl) * 256 + (b + 16 * row)) * 256 + xx;
        if (DIR == 0) x[off + ix] = v[q]; else t[off + it] = v[q];
    }
}
// A side: 8192-element tile = 1024 segments of 8 elements (128 B); segment (tp, kxh) at
// tp*st + kxh*sk + tilebase; other side 32 natural rows of 4 KiB.  DIR 0: T -> rows, 1: rows -> T.
template <int DIR, int LDSB>
__global__ __launch_bounds__(512) void k_aseg(double2* __restrict__ t, double2* __restrict__ x, long tiles, long st, long sk,
                                              long sz, long sb) {
    __shared__ char pad[LDSB > 0 ? LDSB : 1];
    if (LDSB > 0 && threadIdx.x == 0 && tiles < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    const long w = blockIdx.x;
    const size_t off = (size_t)(w / 2048) * N * N * N;
    const int tile = (int)(w % 2048), zh = tile / 16, b = tile % 16;   // zh 0..127
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 512 * q;               // (tp, kxh, in8)
        const int in = i & 7, kxh = (i >> 3) & 63, tp = i >> 9;
        const size_t it = (size_t)tp * st + (size_t)kxh * sk + (size_t)zh * sz + (size_t)b * sb + in;
        const int xx = i & 255, row = (i >> 8) & 15, zl = i >> 12;
        const size_t ix = ((size_t)(zh * 2 + zl) * 256 + (b + 16 * row)) * 256 + xx;
        v[q] = (DIR == 0 || DIR == 2) ? t[off + it] : x[off + ix];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 512 * q;
        const int in = i & 7, kxh = (i >> 3) & 63, tp = i >> 9;
        const size_t it = (size_t)tp * st + (size_t)kxh * sk + (size_t)zh * sz + (size_t)b * sb + in;
        const int xx = i & 255, row = (i >> 8) & 15, zl = i >> 12;
        const size_t ix = ((size_t)(zh * 2 + zl) * 256 + (b + 16 * row)) * 256 + xx;
        if (DIR == 0) x[off + ix] = v[q]; else if (DIR == 1) t[off + it] = v[q]; else x[off + it] = v[q];
    }
}

// T layouts with KL neighbouring kx-groups interleaved below (b, zh):
//   T[tp][kxh / KL][b][zh][kxh % KL][8]
// A tile (fixed b, zh): pieces of KL*128 B; B run (fixed tp, kxh): 128-B pieces at stride KL*128 B.
template <int KL, int DIR>
__global__ __launch_bounds__(512) void k_aseg_kl(double2* __restrict__ t, double2* __restrict__ x, long tiles) {
    __shared__ char pad[140000];
    if (threadIdx.x == 0 && tiles < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    const long w = blockIdx.x;
    const size_t off = (size_t)(w / 2048) * N * N * N;
    const int tile = (int)(w % 2048), zh = tile / 16, b = tile % 16;
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 512 * q;               // (tp, kxh, in8)
        const int in = i & 7, kxh = (i >> 3) & 63, tp = i >> 9;
        const size_t it = ((((size_t)(tp * (64 / KL) + kxh / KL) * 16 + b) * 128 + zh) * KL + kxh % KL) * 8 + in;
        v[q] = t[off + it];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int i = tid + 512 * q;
        const int in = i & 7, kxh = (i >> 3) & 63, tp = i >> 9;
        const size_t it = ((((size_t)(tp * (64 / KL) + kxh / KL) * 16 + b) * 128 + zh) * KL + kxh % KL) * 8 + in;
        x[off + it] = v[q];
    }
}
template <int KL>
__global__ __launch_bounds__(1024) void k_brun_kl(double2* __restrict__ a, long runs) {
    __shared__ char pad[140000];
    if (threadIdx.x == 0 && runs < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    const long w = blockIdx.x;
    const size_t off = (size_t)(w / 1024) * N * N * N;
    const int run = (int)(w % 1024), tp = run >> 6, kxh = run & 63;
    double2 v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {                 // q = b; lane = (zh, in8)
        const int zh = tid >> 3, in = tid & 7;
        const size_t it = ((((size_t)(tp * (64 / KL) + kxh / KL) * 16 + q) * 128 + zh) * KL + kxh % KL) * 8 + in;
        v[q] = a[off + it];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int zh = tid >> 3, in = tid & 7;
        const size_t it = ((((size_t)(tp * (64 / KL) + kxh / KL) * 16 + q) * 128 + zh) * KL + kxh % KL) * 8 + in;
        v[q].x += 1.0;
        a[off + it] = v[q];
    }
}

template <typename F> float timeit(F&& f, int reps = 5) {
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    f(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); for (int i = 0; i < reps; ++i) f(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms / reps;
}

int main(int argc, char** argv) {
    const int streams = argc > 1 ? atoi(argv[1]) : 32;
    const size_t n = (size_t)streams * N * N * N;
    double2 *a, *b;
    CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16));
    CHECK(hipMemset(a, 1, n * 16)); CHECK(hipMemset(b, 0, n * 16));
    const double gb = 2.0 * n * 16 / 1e9;
    printf("bytes moved per launch: %.2f GB\n", gb);
    for (int blocks : {2048, 4096, 16384}) {
        float ms = timeit([&] { k_copy4<<<blocks, 256>>>((const double4*)a, (double4*)b, n / 2); });
        printf("copy double4 grid %5d: %.3f ms  %.0f GB/s\n", blocks, ms, gb / ms * 1e3);
        ms = timeit([&] { k_copy2<<<blocks, 256>>>(a, b, n); });
        printf("copy double2 grid %5d: %.3f ms  %.0f GB/s\n", blocks, ms, gb / ms * 1e3);
    }
    const int ntiles = N * N / 16;
    const long total = (long)ntiles * streams;
#define RUN(AX, L, G) { float ms = timeit([&] { k_tile<AX, L><<<G, 256>>>(a, b, ntiles, total); }); \
        printf("tile ax=%d lds=%6d grid=%7ld: %.3f ms  %.0f GB/s\n", AX, L, (long)(G), ms, gb / ms * 1e3); }
    RUN(0, 0, total) RUN(1, 0, total) RUN(2, 0, total)
    RUN(0, 73984, total) RUN(1, 73984, total) RUN(2, 73984, total)
    RUN(0, 73984, 512) RUN(1, 73984, 512) RUN(2, 73984, 512)
    RUN(0, 39000, 1024) RUN(1, 39000, 1024) RUN(2, 39000, 1024)
    RUN(0, 0, 2048) RUN(1, 0, 2048) RUN(2, 0, 2048)
    // working-set sweep: in-place y-pattern tile pass over buffers of growing size (Infinity Cache reach)
    for (double mib : {32.0, 64.0, 128.0, 192.0, 256.0, 384.0, 512.0, 1024.0, 4096.0}) {
        const long tiles = (long)(mib * 1048576.0 / 65536.0);      // one tile = 64 KiB
        const long tot = tiles < total ? tiles : total;
        const double g2 = 2.0 * tot * 65536.0 / 1e9;
        float ms = timeit([&] { k_tile<1, 73984><<<tot, 256>>>(a, a, ntiles, tot); }, 20);
        printf("in-place ax=1 working set %6.0f MiB: %.4f ms  %.0f GB/s\n", mib, ms, g2 / ms * 1e3);
        ms = timeit([&] { k_tile<2, 73984><<<tot, 256>>>(a, a, ntiles, tot); }, 20);
        printf("in-place ax=2 working set %6.0f MiB: %.4f ms  %.0f GB/s\n", mib, ms, g2 / ms * 1e3);
    }
    {
        const long tb = 1024L * streams;
        float ms = timeit([&] { k_btile<140000><<<tb, 512>>>(a, tb); });
        printf("B tile 16K elems, 512 thr, 1 WG/CU, grid %ld: %.3f ms  %.0f GB/s\n", tb, ms, gb / ms * 1e3);
        ms = timeit([&] { k_btile<0><<<tb, 512>>>(a, tb); });
        printf("B tile 16K elems, 512 thr, no LDS limit     : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { k_btile1024<140000><<<tb, 1024>>>(a, tb); });
        printf("B tile 16K elems, 1024 thr, 1 WG/CU          : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { k_atile_loop<0, 73984><<<tb, 256>>>(a, b, tb); });
        printf("A tile loop, read 64-B segs, 2 WG/CU         : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { k_atile_loop<1, 73984><<<tb, 256>>>(a, b, tb); });
        printf("A tile loop, write 64-B segs, 2 WG/CU        : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { k_atile_loop<0, 0><<<tb, 256>>>(a, b, tb); });
        printf("A tile loop, read 64-B segs, no LDS limit    : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { k_atile_loop<1, 0><<<tb, 256>>>(a, b, tb); });
        printf("A tile loop, write 64-B segs, no LDS limit   : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        {
            const long runs = 1024L * streams;
            ms = timeit([&] { k_bhalf<1, 73984><<<2 * runs, 512>>>(a, runs); });
            printf("B half-tiles (64-B chunks), partner +1, 2 WG/CU: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
            ms = timeit([&] { k_bhalf<8, 73984><<<2 * runs, 512>>>(a, runs); });
            printf("B half-tiles (64-B chunks), partner +8, 2 WG/CU: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
            ms = timeit([&] { k_bhalf<16, 73984><<<2 * runs, 512>>>(a, runs); });
            printf("B half-tiles (64-B chunks), partner +16, 2 WG/CU: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
            ms = timeit([&] { k_bhalf32<8, 73984><<<2 * runs, 512>>>(a, runs); });
            printf("B half-tiles (2x32-B pieces), partner +8, 2 WG/CU: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
            {
                const long tl2 = 2 * 2048L * streams;
                const long st = 64L * 128 * 16 * 8, sk = 128L * 16 * 8, sz = 16 * 8, sb = 8;
                ms = timeit([&] { k_aseg64<0, 8, 73984><<<tl2, 256>>>(a, b, tl2, st, sk, sz, sb); });
                printf("A 4K tile (64-B chunks, partner +8), T -> rows: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_aseg64<1, 8, 73984><<<tl2, 256>>>(a, b, tl2, st, sk, sz, sb); });
                printf("A 4K tile (64-B chunks, partner +8), rows -> T: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
            }
            const long tl = 2048L * streams;
            struct L { const char* name; long st, sk, sz, sb; } ls[] = {
                {"[tp][kxh][zh][b]  (B contiguous)", 64L * 128 * 16 * 8, 128L * 16 * 8, 16 * 8, 8},
                {"[tp][zh][kxh][b]", 128L * 64 * 16 * 8, 16 * 8, 64L * 16 * 8, 8},
                {"[tp][zh][b][kxh]  (A contiguous)", 128L * 16 * 64 * 8, 8, 16L * 64 * 8, 64 * 8},
                {"[zh][b][tp][kxh]  (A tile contiguous)", 64 * 8, 8, 16L * 16 * 64 * 8, 16L * 64 * 8},
            };
            {
                auto& l = ls[0];
                ms = timeit([&] { k_aseg<2, 73984><<<tl, 512>>>(a, b, tl, l.st, l.sk, l.sz, l.sb); });
                printf("A 8K tile, T -> T (both 128-B segments), 2 WG/CU: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_aseg<2, 140000><<<tl, 512>>>(a, b, tl, l.st, l.sk, l.sz, l.sb); });
                printf("A 8K tile, T -> T (both 128-B segments), 1 WG/CU: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
            }
            {
                const long tb2 = 1024L * streams;
                ms = timeit([&] { k_aseg_kl<1, 0><<<tl, 512>>>(a, b, tl); });
                printf("KL=1  A T->T : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_brun_kl<1><<<tb2, 1024>>>(a, tb2); });
                printf("KL=1  B run  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_aseg_kl<2, 0><<<tl, 512>>>(a, b, tl); });
                printf("KL=2  A T->T : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_brun_kl<2><<<tb2, 1024>>>(a, tb2); });
                printf("KL=2  B run  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_aseg_kl<4, 0><<<tl, 512>>>(a, b, tl); });
                printf("KL=4  A T->T : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_brun_kl<4><<<tb2, 1024>>>(a, tb2); });
                printf("KL=4  B run  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_aseg_kl<8, 0><<<tl, 512>>>(a, b, tl); });
                printf("KL=8  A T->T : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_brun_kl<8><<<tb2, 1024>>>(a, tb2); });
                printf("KL=8  B run  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_aseg_kl<64, 0><<<tl, 512>>>(a, b, tl); });
                printf("KL=64 A T->T : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
                ms = timeit([&] { k_brun_kl<64><<<tb2, 1024>>>(a, tb2); });
                printf("KL=64 B run  : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
            }
            for (auto& l : ls) {
                ms = timeit([&] { k_aseg<0, 73984><<<tl, 512>>>(a, b, tl, l.st, l.sk, l.sz, l.sb); });
                printf("A 8K tile, T%s -> rows: %.3f ms  %.0f GB/s\n", l.name, ms, gb / ms * 1e3);
                ms = timeit([&] { k_aseg<1, 73984><<<tl, 512>>>(a, b, tl, l.st, l.sk, l.sz, l.sb); });
                printf("A 8K tile, rows -> T%s: %.3f ms  %.0f GB/s\n", l.name, ms, gb / ms * 1e3);
            }
        }
        const long ta1 = 1024L * 4 * streams, ta2 = 1024L * 2 * streams, ta4 = 1024L * streams;
        ms = timeit([&] { k_atile<1, 256, 73984><<<ta1, 256>>>(a, b, ta1); });
        printf("A tile 64-B segments, 256 thr, 2 WG/CU       : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { k_atile<2, 512, 73984><<<ta2, 512>>>(a, b, ta2); });
        printf("A tile 128-B segments, 512 thr, 2 WG/CU      : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { k_atile<2, 512, 140000><<<ta2, 512>>>(a, b, ta2); });
        printf("A tile 128-B segments, 512 thr, 1 WG/CU      : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
        ms = timeit([&] { k_atile<4, 1024, 140000><<<ta4, 1024>>>(a, b, ta4); });
        printf("A tile 256-B segments, 1024 thr, 1 WG/CU     : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
    }
    // in place
    { float ms = timeit([&] { k_tile<1, 73984><<<total, 256>>>(a, a, ntiles, total); });
      printf("tile ax=1 in place lds=73984: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3); }
    return 0;
}
