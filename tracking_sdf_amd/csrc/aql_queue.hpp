// aql_queue.hpp -- a user-mode HSA queue of the library's own for the tracker's Gauss-Newton passes >= 1.
//
// A pass is one launch whose result the host waits for before it can launch the next one; hipLaunchKernel spends 2.5-2.8 us
// of host time per launch on that critical path, a hand-written AQL dispatch 0.2 (tools/aql_probe.hip, profiles/
// r05_aql_probe.json: 14.7 -> 12.3 us from submission to the host seeing the result; submit() adds the 0.9 us read-back of the
// arguments that makes the doorbell safe: r05_aql_readback.json).  The kernel is the SAME device code:
// the build also emits tsdf_kernels.hip as a stand-alone code object (lib/tsdf_kernels.hsaco), loaded here through HSA.
// Anything that goes wrong while setting this up (no code object, symbol, queue ...) just leaves the HIP launch in place.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <string>

namespace tsdf {

class AqlQueue {
public:
    // hsaco_path: the code object; symbol_prefix: the kernel's mangled name starts with it; explicit_bytes: size of the
    // kernel's explicit arguments (checked against the code object: kernarg segment = explicit + 256 hidden bytes)
    bool init(int hip_device, const char* hsaco_path, const char* symbol_prefix, size_t explicit_bytes, std::string* err);
    bool ready() const { return queue_ != nullptr; }
    // One dispatch of `workgroups` x `block` threads (1-D) with the given explicit arguments.  Packets of this queue run in
    // order (barrier bit); acquire / release fences at agent scope, as HIP's own kernel packets have.  Returns false when the
    // ring is full (cannot happen with host-synchronous passes) -- the caller then launches through HIP.
    bool submit(const void* explicit_args, uint32_t workgroups, uint32_t block);
    void wait_idle();          // until the last submitted packet has completed
    void destroy();
    ~AqlQueue() { destroy(); }

private:
    void* queue_ = nullptr;            // hsa_queue_t*
    uint64_t kernel_object_ = 0;
    uint32_t group_bytes_ = 0, private_bytes_ = 0, kernarg_bytes_ = 0;
    size_t explicit_bytes_ = 0;
    char* kernarg_ = nullptr;          // a ring of buffers in device memory the host writes through the BAR
    size_t kernarg_stride_ = 0;
    uint64_t submitted_ = 0;
    uint64_t signal_ = 0;              // hsa_signal_t handle: completion of the last packet
    uint64_t executable_ = 0, reader_ = 0;
    bool hsa_up_ = false;
    bool readback_ = true;             // read a byte of the arguments back before the doorbell (TSDF_AQL_READBACK=0: not)
    unsigned readback_sink_ = 0;
};

}  // namespace tsdf
