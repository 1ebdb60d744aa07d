// Run-time binding of RCCL (see comm.hpp).
#include "comm.hpp"

#include <dlfcn.h>

#include <cstdlib>

#include <mutex>

namespace msm {
namespace {
Rccl g_rccl;
bool g_tried = false;
std::string g_err;
std::mutex g_mutex;

void* open_rccl(std::string& origin) {
    // MSM_RCCL_LIB names the one library to bind (a site's own build; the tests name a file that does not exist to
    // take the failure path); otherwise a copy that is already mapped wins (RTLD_NOLOAD), then the loader's search
    // path, then the ROCm tree
    if (const char* forced = std::getenv("MSM_RCCL_LIB")) {
        origin = forced;
        return dlopen(forced, RTLD_NOW | RTLD_LOCAL);
    }
    static const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names)
        if (void* h = dlopen(n, RTLD_NOW | RTLD_NOLOAD)) { origin = std::string(n) + " (already loaded)"; return h; }
    for (const char* n : names)
        if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) { origin = n; return h; }
    static const char* paths[] = {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    for (const char* n : paths)
        if (void* h = dlopen(n, RTLD_NOW | RTLD_LOCAL)) { origin = n; return h; }
    return nullptr;
}
}  // namespace

const Rccl* rccl(std::string& err) {
    std::lock_guard<std::mutex> lk(g_mutex);
    if (!g_tried) {
        g_tried = true;
        void* h = open_rccl(g_rccl.origin);
        if (!h) {
            const char* why = dlerror();      // one call: it hands the message over and clears it
            g_err = std::string("librccl not found: ") + (why ? why : "dlopen failed");
        } else {
            auto sym = [&](const char* name) -> void* {
                void* p = dlsym(h, name);
                if (!p && g_err.empty()) g_err = std::string("librccl lacks ") + name;
                return p;
            };
            g_rccl.get_unique_id = (decltype(g_rccl.get_unique_id))sym("ncclGetUniqueId");
            g_rccl.comm_init_rank = (decltype(g_rccl.comm_init_rank))sym("ncclCommInitRank");
            g_rccl.comm_destroy = (decltype(g_rccl.comm_destroy))sym("ncclCommDestroy");
            g_rccl.all_reduce = (decltype(g_rccl.all_reduce))sym("ncclAllReduce");
            g_rccl.error_string = (decltype(g_rccl.error_string))sym("ncclGetErrorString");
        }
    }
    if (!g_err.empty()) { err = g_err; return nullptr; }
    return &g_rccl;
}

}  // namespace msm
