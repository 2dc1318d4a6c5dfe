// aql_queue.hpp -- a user-mode HSA queue of the library's own for the tracker's Gauss-Newton passes >= 1.  OPT-IN (TSDF_AQL=1).
//
// A pass is one launch whose result the host waits for before it can launch the next one; hipLaunchKernel spends 2.5-2.8 us
// of host time per launch on that critical path, a hand-written AQL dispatch 0.2 (tools/aql_probe.hip, profiles/
// r06_pass_floor.jsonl: 2.3-3.3 us less from submission to the host seeing the result at every grid shape; submit() adds the
// 0.9 us read-back of the arguments that makes the doorbell safe).  Inside the frame loop that came to 0.3 us per pass
// (+0.1-0.6 % frames/s, profiles/r05_aql_readback.json) -- not enough to pay for a second submission path, so the default
// since round 6 is the HIP stream and this queue is an option.  The kernel is the SAME device code: the build also emits
// track_kernels.hip as a stand-alone code object (lib/tsdf_track.hsaco) carrying the library's build id, loaded here through HSA.
// Anything that goes wrong while setting this up (no code object, another build's code object, no queue ...) just leaves
// the HIP launch in place.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include <string>

namespace tsdf {

class AqlQueue {
public:
    // hsaco_path: the code object; symbol_prefix: the kernel's mangled name starts with it; explicit_bytes: size of the
    // kernel's explicit arguments (checked against the code object: kernarg segment = explicit + 256 hidden bytes);
    // build_id: what the code object's `tsdf_track_build_id` variable must hold (the library's own TSDF_BUILD_ID)
    bool init(int hip_device, const char* hsaco_path, const char* symbol_prefix, size_t explicit_bytes, const char* build_id, std::string* err);
    bool ready() const { return queue_ != nullptr && !dead_; }
    // One dispatch of `workgroups` x `block` threads (1-D) with the given explicit arguments.  Packets of this queue run in
    // order (barrier bit); acquire / release fences at agent scope, as HIP's own kernel packets have.  Returns false -- the
    // caller then launches through HIP -- when the queue is dead (a fault, or a wait that timed out) or when as many packets
    // are outstanding as there are argument buffers (cannot happen with host-synchronous passes).
    bool submit(const void* explicit_args, uint32_t workgroups, uint32_t block);
    // until the last submitted packet has completed.  False: it did not within 2 s, or the queue reported a fault -- the queue
    // is dead from then on (ready() is false, submit() refuses) and destroy() leaks what a running packet could still touch.
    bool wait_idle();
    bool dead() const { return dead_; }
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
    unsigned readback_sink_ = 0;
    bool dead_ = false;
    static void on_queue_error(int status, void* queue, void* self);     // hsa_queue_create's callback: records the fault
};

}  // namespace tsdf
