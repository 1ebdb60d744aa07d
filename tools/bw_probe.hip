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
    // in place
    { float ms = timeit([&] { k_tile<1, 73984><<<total, 256>>>(a, a, ntiles, total); });
      printf("tile ax=1 in place lds=73984: %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3); }
    return 0;
}
