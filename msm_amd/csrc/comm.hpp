// RCCL, bound at run time.  The library has no link-time dependency on librccl: a process that never calls
// msm_comm_* does not load it, and one that already carries an RCCL (a PyTorch process) reuses that copy instead of
// bringing a second one into the address space.
#pragma once
#include <cstddef>
#include <string>

#include <hip/hip_runtime.h>

namespace msm {

constexpr int COMM_ID_BYTES = 128;            // sizeof(ncclUniqueId)
struct CommId { char bytes[COMM_ID_BYTES]; };

enum CommOp { COMM_SUM = 0, COMM_MAX = 2, COMM_MIN = 3 };                 // ncclRedOp_t values
enum CommType { COMM_I64 = 4, COMM_F64 = 8 };                             // ncclDataType_t values

struct Rccl {
    // all return 0 (ncclSuccess) on success
    int (*get_unique_id)(CommId*) = nullptr;
    int (*comm_init_rank)(void** comm, int nranks, CommId id, int rank) = nullptr;
    int (*comm_destroy)(void* comm) = nullptr;
    int (*all_reduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) = nullptr;
    const char* (*error_string)(int) = nullptr;
    std::string origin;                        // which shared object the symbols came from
};

// Loads the symbols once; null + err when no RCCL can be found.
const Rccl* rccl(std::string& err);

}  // namespace msm
