// Gather probe for the A tiles of the 512^3 single-precision tiling (measurement tool, no arithmetic).
// A tile (b, zh) takes one 64-byte piece [zl 4][kx & 1] per (t', kx >> 1) out of T[t'][kx >> 1][b][z][kx & 1]: 4096 pieces
// of 64 B, 256 KB apart along kx >> 1 and 64 MiB apart along t'.  512 threads, one workgroup per CU (LDS padding), every
// thread 32 loads of 16 B, all issued before the first use.  Variants of who loads what:
//   0  as the kernels do today: the two lanes of a piece take 16 B each at +0 / +16, a second instruction the same at +32
//      (a wave instruction touches 32 pieces, 32 B of each)
//   1  four lanes take one piece contiguously (64 B); the second instruction takes the quad's other piece
//      (a wave instruction touches 16 pieces, 64 B of each)
//   2  eight lanes take a whole 128-byte line (the piece and its partner tile's): twice the bytes, for reference
// Build: hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o build/gather_probe ; run: build/gather_probe [streams]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

typedef float f4 __attribute__((ext_vector_type(4)));
constexpr size_t T_ZH = 8, T_B = 128 * T_ZH, T_KXH = 32 * T_B, T_TP = 256 * T_KXH;   // in complex64 elements (8 B)
constexpr size_t GRID = 512ul * 512 * 512;

// 32-byte pieces: the layout T2[t'][kx][b][z] a tiling with pass-B runs of 2^14 points would have (a run = one kx: a
// contiguous 128 KB block; an A tile's piece = its four z of one kx).  Four tiles (a z quad of quads) share a 128-byte
// line and take neighbouring slots of one XCD.  WRITE = 1 stores instead of loading.
template <int WRITE, int PIECE64>
__global__ __launch_bounds__(512) void k_piece(float2* __restrict__ T, float* __restrict__ out, int ntiles_total) {
    __shared__ char pad[140 * 1024];
    if (threadIdx.x == 0 && ntiles_total < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    const int id = blockIdx.x & 4095, stream = blockIdx.x >> 12;
    const int xcd = id & 7, slot = id >> 3;
    float2* base = T + (size_t)stream * GRID;
    f4 v[32];
    size_t off, kstride = T_TP;
    int second;                      // offset of the thread's second 16-byte access, in f4 units
    if (PIECE64) {                   // today's layout, two lanes per 64-byte piece
        const int group = ((slot >> 1) << 3) | xcd;
        const int zh = ((group & 63) << 1) | (slot & 1), b = group >> 6, kxh = tid >> 1;
        off = (size_t)kxh * T_KXH + b * T_B + (size_t)((zh + 4 * kxh) & 127) * T_ZH + 2 * (tid & 1);
        second = 2;
    } else {                         // T2: one lane per 32-byte piece
        const int group = ((slot >> 2) << 3) | xcd;
        const int zh = ((group & 31) << 2) | (slot & 3), b = group >> 5;
        off = (size_t)tid * (32 * 512) + (size_t)b * 512 + (size_t)((4 * zh + 16 * tid) & 511);   // (z rotated per kx)
        second = 1;
    }
    if (WRITE) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            f4 d = {(float)tid, (float)k, 1.0f, 2.0f};
            f4* p = (f4*)(base + off + (size_t)k * kstride);
            p[0] = d;
            p[second] = d;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f4* p = (const f4*)(base + off + (size_t)k * kstride);
            v[2 * k] = p[0];
            v[2 * k + 1] = p[second];
        }
        f4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += v[i];
        if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;
    }
}

template <int V>
__global__ __launch_bounds__(512) void k_gather(const float2* __restrict__ T, float* __restrict__ out, int ntiles_total) {
    __shared__ char pad[140 * 1024];
    if (threadIdx.x == 0 && ntiles_total < 0) pad[0] = 1;
    const int tid = threadIdx.x;
    const int id = blockIdx.x & 4095, stream = blockIdx.x >> 12;
    const int xcd = id & 7, slot = id >> 3;
    const int group = ((slot >> 1) << 3) | xcd;
    const int zh = ((group & 63) << 1) | (slot & 1), b = group >> 6;
    const float2* base = T + (size_t)stream * GRID;
    f4 acc = {0, 0, 0, 0};
    f4 v[32];
    if (V == 0) {
        const int kxh = tid >> 1;
        const size_t off = (size_t)kxh * T_KXH + b * T_B + (size_t)((zh + 4 * kxh) & 127) * T_ZH + 2 * (tid & 1);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const f4* p = (const f4*)(base + off + (size_t)k * T_TP);
            v[2 * k] = p[0];
            v[2 * k + 1] = p[2];
        }
    } else if (V == 1) {
        const int quad = tid >> 2, j = tid & 3;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int kxh = 2 * quad + h;
                const size_t off = (size_t)kxh * T_KXH + b * T_B + (size_t)((zh + 4 * kxh) & 127) * T_ZH + 2 * j;
                v[2 * k + h] = *(const f4*)(base + off + (size_t)k * T_TP);
            }
        }
    } else {
        // eight lanes per 128-byte line; 64 threads x 8 = the 256 kx >> 1 of one t' per two instructions
        const int oct = tid >> 3, j = tid & 7;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int kxh = 4 * oct + h;
                const size_t off = (size_t)kxh * T_KXH + b * T_B + (size_t)(((zh & ~1) + 4 * kxh) & 127) * T_ZH + 2 * j;
                const f4 d = *(const f4*)(base + off + (size_t)k * T_TP);
                if (h < 2) v[2 * k + h] = d; else v[2 * k + h - 2] += d;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += v[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[blockIdx.x] = acc.x;   // never true: keeps the loads
}

int main(int argc, char** argv) {
    const int streams = argc > 1 ? atoi(argv[1]) : 4;
    float2* T; float* out;
    CHECK(hipMalloc(&T, GRID * sizeof(float2) * streams));
    CHECK(hipMemset(T, 0, GRID * sizeof(float2) * streams));
    CHECK(hipMalloc(&out, 4096 * streams * sizeof(float)));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int v = 0; v < 3; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (v == 0) k_gather<0><<<4096 * streams, 512>>>(T, out, 4096 * streams);
            else if (v == 1) k_gather<1><<<4096 * streams, 512>>>(T, out, 4096 * streams);
            else k_gather<2><<<4096 * streams, 512>>>(T, out, 4096 * streams);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        const double useful = (double)GRID * 8 * streams;
        printf("variant %d: %.3f ms  %.2f TB/s of tile bytes%s\n", v, best, useful / best / 1e9, v == 2 ? " (moves twice that)" : "");
    }
    const char* names[4] = {"read 64-B pieces (today)", "read 32-B pieces (T2)", "write 64-B pieces (today)", "write 32-B pieces (T2)"};
    for (int v = 0; v < 4; ++v) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            CHECK(hipEventRecord(e0));
            if (v == 0) k_piece<0, 1><<<4096 * streams, 512>>>(T, out, 4096 * streams);
            else if (v == 1) k_piece<0, 0><<<4096 * streams, 512>>>(T, out, 4096 * streams);
            else if (v == 2) k_piece<1, 1><<<4096 * streams, 512>>>(T, out, 4096 * streams);
            else k_piece<1, 0><<<4096 * streams, 512>>>(T, out, 4096 * streams);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0 && ms < best) best = ms;
        }
        printf("%s: %.3f ms  %.2f TB/s of tile bytes\n", names[v], best, (double)GRID * 8 * streams / best / 1e9);
    }
    CHECK(hipGetLastError());
    return 0;
}
