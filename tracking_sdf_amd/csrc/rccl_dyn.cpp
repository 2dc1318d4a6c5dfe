// rccl_dyn.cpp -- see rccl_dyn.hpp.
#include "rccl_dyn.hpp"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <mutex>

namespace tsdf {
namespace rccl {
namespace {

// The few RCCL entry points used, with the ABI of rccl.h (ncclUniqueId = 128 opaque bytes,
// ncclFloat64 = 8, ncclSum = 0, ncclSuccess = 0).
struct UniqueId { char internal[128]; };
using GetUniqueIdFn = int (*)(UniqueId*);
using CommInitRankFn = int (*)(void** comm, int nranks, UniqueId id, int rank);
using AllReduceFn = int (*)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s);
using CommDestroyFn = int (*)(void* comm);
using GetErrorStringFn = const char* (*)(int);

struct Api {
    void* lib = nullptr;
    GetUniqueIdFn get_unique_id = nullptr;
    CommInitRankFn comm_init_rank = nullptr;
    AllReduceFn all_reduce = nullptr;
    CommDestroyFn comm_destroy = nullptr;
    GetErrorStringFn get_error_string = nullptr;
    std::string load_error;
};

Api& api() {
    static Api a;
    static std::once_flag once;
    std::call_once(once, [] {
        // Prefer an RCCL the process already holds (torch ships its own), then the system one.
        // TSDF_RCCL_LIBRARY names the library outright (a deployment whose RCCL lives elsewhere; the test-suite points
        // it at a file that does not exist to exercise the failure path).
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        const char* forced = std::getenv("TSDF_RCCL_LIBRARY");
        std::string why = "?";
        auto try_open = [&](const char* name, int flags) {
            void* l = dlopen(name, flags);
            if (!l) { const char* e = dlerror(); if (e) why = e; }    // dlerror() clears the message: read it ONCE
            return l;
        };
        if (forced && *forced) {
            a.lib = try_open(forced, RTLD_NOW | RTLD_LOCAL);
        } else {
            a.lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_NOLOAD);
            for (size_t i = 0; !a.lib && i < sizeof(names) / sizeof(names[0]); ++i) a.lib = try_open(names[i], RTLD_NOW | RTLD_LOCAL);
        }
        if (!a.lib) { a.load_error = "cannot load librccl: " + why; return; }
        a.get_unique_id = (GetUniqueIdFn)dlsym(a.lib, "ncclGetUniqueId");
        a.comm_init_rank = (CommInitRankFn)dlsym(a.lib, "ncclCommInitRank");
        a.all_reduce = (AllReduceFn)dlsym(a.lib, "ncclAllReduce");
        a.comm_destroy = (CommDestroyFn)dlsym(a.lib, "ncclCommDestroy");
        a.get_error_string = (GetErrorStringFn)dlsym(a.lib, "ncclGetErrorString");
        if (!a.get_unique_id || !a.comm_init_rank || !a.all_reduce || !a.comm_destroy) {
            a.load_error = "librccl is missing an expected symbol";
            a.lib = nullptr;
        }
    });
    return a;
}

std::string describe(int rc) {
    Api& a = api();
    if (a.get_error_string) return a.get_error_string(rc);
    return "nccl error " + std::to_string(rc);
}

}  // namespace

bool unique_id(void* id128, std::string* err) {
    Api& a = api();
    if (!a.lib) { if (err) *err = a.load_error; return false; }
    UniqueId id;
    const int rc = a.get_unique_id(&id);
    if (rc != 0) { if (err) *err = describe(rc); return false; }
    std::memcpy(id128, &id, sizeof id);
    return true;
}

bool Comm::init(int nranks, int rank, const void* id128, std::string* err) {
    Api& a = api();
    if (!a.lib) { if (err) *err = a.load_error; return false; }
    destroy();
    UniqueId id;
    std::memcpy(&id, id128, sizeof id);
    void* c = nullptr;
    const int rc = a.comm_init_rank(&c, nranks, id, rank);
    if (rc != 0 || !c) { if (err) *err = describe(rc); return false; }
    comm_ = c;
    nranks_ = nranks;
    return true;
}

bool Comm::allreduce_sum_f64(double* dev_buf, int n, hipStream_t stream, std::string* err) {
    Api& a = api();
    if (!comm_) { if (err) *err = "communicator not initialised"; return false; }
    const int rc = a.all_reduce(dev_buf, dev_buf, (size_t)n, /*ncclFloat64*/ 8, /*ncclSum*/ 0, comm_, stream);
    if (rc != 0) { if (err) *err = describe(rc); return false; }
    return true;
}

void Comm::destroy() {
    if (comm_) {
        Api& a = api();
        if (a.comm_destroy) (void)a.comm_destroy(comm_);
        comm_ = nullptr;
        nranks_ = 1;
    }
}

}  // namespace rccl
}  // namespace tsdf
