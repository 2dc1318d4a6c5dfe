// rccl_dyn.hpp -- RCCL bound at run time (dlopen), so libtsdf_hip.so loads on a box without RCCL
// and shares the RCCL instance a host process (e.g. torch.distributed) may already have loaded.
// Used for exactly one collective: the per-Gauss-Newton-iteration sum of the 6x6/6 normal
// equations over the x-slab ranks, a 240-byte all-reduce over xGMI.
#pragma once

#include <hip/hip_runtime_api.h>

#include <string>

namespace tsdf {
namespace rccl {

bool unique_id(void* id128, std::string* err);

class Comm {
public:
    bool active() const { return comm_ != nullptr; }
    bool init(int nranks, int rank, const void* id128, std::string* err);
    bool allreduce_sum_f64(double* dev_buf, int n, hipStream_t stream, std::string* err);
    void destroy();
    int nranks() const { return nranks_; }

private:
    void* comm_ = nullptr;
    int nranks_ = 1;
};

}  // namespace rccl
}  // namespace tsdf
