// A stand-in for librccl for tests on a box with ONE GPU (test infrastructure, never shipped or measured).
//
// RCCL refuses a communicator whose ranks share a GPU, so on a one-GPU box the library's multi-rank code (msm_comm_*,
// the per-step density all-reduce, the synthesizer's all-reduce, the drivers' flag exchanges) can only ever meet a
// communicator of one.  This library exports the five entry points libmsm_hip.so binds (MSM_RCCL_LIB names it) and
// carries the collectives over POSIX shared memory between processes of one host, so that two or more ranks can run
// against one GPU and the ORDER of the library's collective calls is exercised for real: a rank that issues one
// collective more, fewer or in another order than its peers ends in a timeout here, as it would hang in RCCL.
//
// all-reduce: synchronous.  The caller's stream is drained, the send buffer is copied to this rank's slot of the
// shared segment in chunks, the ranks meet at a barrier, every rank reduces all slots in rank order (bitwise the same
// result everywhere), a second barrier frees the slots, the result is copied to the receive buffer.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16;
constexpr size_t CHUNK = (size_t)4 << 20;        // bytes per rank and round
constexpr int TIMEOUT_S = 120;

struct Header {
    std::atomic<int> arrived;                    // ranks that have mapped the segment
    std::atomic<unsigned> gate;                  // barrier: arrivals of the current round
    std::atomic<unsigned> round;                 // barrier: completed rounds
    std::atomic<int> left;                       // ranks that have destroyed their communicator
    std::atomic<unsigned long long> calls[MAX_RANKS];   // collectives issued per rank (diagnostics)
};

struct Comm {
    char name[128];
    int rank, world;
    size_t bytes;
    Header* hdr;
    unsigned char* slots;
    std::vector<unsigned char> local;
};

bool barrier(Comm* c, const char* what) {
    const unsigned my_round = c->hdr->round.load();
    if (c->hdr->gate.fetch_add(1) + 1 == (unsigned)c->world) {
        c->hdr->gate.store(0);
        c->hdr->round.fetch_add(1);
        return true;
    }
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->round.load() == my_round) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(TIMEOUT_S)) {
            std::fprintf(stderr, "fake_rccl: rank %d waited %d s at the barrier of %s; collectives issued per rank:", c->rank, TIMEOUT_S, what);
            for (int r = 0; r < c->world; ++r) std::fprintf(stderr, " %llu", c->hdr->calls[r].load());
            std::fprintf(stderr, " -- the ranks disagree on the order or number of collective calls\n");
            return false;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    return true;
}

template <typename T>
void reduce_into(T* acc, const T* in, size_t n, int op) {
    for (size_t i = 0; i < n; ++i) {
        switch (op) {
            case 0: acc[i] = acc[i] + in[i]; break;                  // ncclSum
            case 1: acc[i] = acc[i] * in[i]; break;                  // ncclProd
            case 2: acc[i] = in[i] > acc[i] ? in[i] : acc[i]; break; // ncclMax
            default: acc[i] = in[i] < acc[i] ? in[i] : acc[i]; break; // ncclMin
        }
    }
}

}  // namespace

extern "C" {

int ncclGetUniqueId(void* id) {
    char* p = (char*)id;
    std::memset(p, 0, 128);
    const auto ns = std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
    std::snprintf(p, 128, "/msm-fake-rccl-%d-%lld", (int)getpid(), (long long)ns);
    return 0;
}

struct UniqueId { char bytes[128]; };

int ncclCommInitRank(void** comm, int nranks, UniqueId id, int rank) {
    if (!comm || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.bytes[0] != '/') return 4;   // ncclInvalidArgument
    Comm* c = new Comm();
    std::memcpy(c->name, id.bytes, 128);
    c->name[127] = 0;
    c->rank = rank; c->world = nranks;
    c->bytes = sizeof(Header) + (size_t)nranks * CHUNK;
    const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
    if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) { std::perror("fake_rccl: shm_open"); delete c; return 2; }   // ncclSystemError
    void* m = mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (m == MAP_FAILED) { std::perror("fake_rccl: mmap"); delete c; return 2; }
    c->hdr = (Header*)m;                                   // a fresh segment is zero-filled: all counters start at 0
    c->slots = (unsigned char*)m + sizeof(Header);
    c->local.resize(CHUNK);
    c->hdr->arrived.fetch_add(1);
    const auto t0 = std::chrono::steady_clock::now();
    while (c->hdr->arrived.load() < nranks) {
        if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(TIMEOUT_S)) {
            std::fprintf(stderr, "fake_rccl: rank %d: only %d of %d ranks joined the communicator\n", rank, c->hdr->arrived.load(), nranks);
            munmap(m, c->bytes);
            delete c;
            return 2;
        }
        std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    *comm = c;
    return 0;
}

int ncclCommDestroy(void* comm) {
    Comm* c = (Comm*)comm;
    if (!c) return 0;
    if (c->hdr->left.fetch_add(1) + 1 == c->world) shm_unlink(c->name);
    munmap((void*)c->hdr, c->bytes);
    delete c;
    return 0;
}

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t stream) {
    Comm* c = (Comm*)comm;
    if (!c || !send || !recv) return 4;
    const size_t esz = dtype == 8 ? 8 : dtype == 7 ? 4 : dtype == 4 ? 8 : 0;     // ncclFloat64, ncclFloat32, ncclInt64
    if (esz == 0 || op < 0 || op > 3) return 4;
    c->hdr->calls[c->rank].fetch_add(1);
    if (hipStreamSynchronize(stream) != hipSuccess) return 1;                     // ncclUnhandledCudaError
    const size_t per = CHUNK / esz;
    for (size_t off = 0; off < count; off += per) {
        const size_t n = count - off < per ? count - off : per;
        unsigned char* mine = c->slots + (size_t)c->rank * CHUNK;
        if (hipMemcpy(mine, (const unsigned char*)send + off * esz, n * esz, hipMemcpyDeviceToHost) != hipSuccess) return 1;
        if (!barrier(c, "an all-reduce (slots filled)")) return 3;                // ncclInternalError
        std::memcpy(c->local.data(), c->slots, n * esz);
        for (int r = 1; r < c->world; ++r) {
            const unsigned char* in = c->slots + (size_t)r * CHUNK;
            if (dtype == 8) reduce_into((double*)c->local.data(), (const double*)in, n, op);
            else if (dtype == 7) reduce_into((float*)c->local.data(), (const float*)in, n, op);
            else reduce_into((long long*)c->local.data(), (const long long*)in, n, op);
        }
        if (!barrier(c, "an all-reduce (slots read)")) return 3;
        if (hipMemcpy((unsigned char*)recv + off * esz, c->local.data(), n * esz, hipMemcpyHostToDevice) != hipSuccess) return 1;
    }
    return 0;
}

const char* ncclGetErrorString(int code) {
    switch (code) {
        case 0: return "no error";
        case 1: return "unhandled HIP error (fake_rccl)";
        case 2: return "system error (fake_rccl: shared memory)";
        case 3: return "internal error (fake_rccl: a rank did not arrive; see stderr)";
        case 4: return "invalid argument (fake_rccl)";
        default: return "unknown error (fake_rccl)";
    }
}

}  // extern "C"
