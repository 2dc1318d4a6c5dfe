// aql_queue.cpp -- see aql_queue.hpp.  Host code only (HSA runtime API); gfx950 / code object v5.
#include "aql_queue.hpp"

#include <hip/hip_runtime_api.h>
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace tsdf {
namespace {

struct AgentPick { uint32_t want_bdf, want_domain; bool any; hsa_agent_t gpu; bool have_gpu; hsa_agent_t cpu; bool have_cpu; };
hsa_status_t pick_agent(hsa_agent_t a, void* data) {
    AgentPick* p = static_cast<AgentPick*>(data);
    hsa_device_type_t t;
    if (hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t) != HSA_STATUS_SUCCESS) return HSA_STATUS_SUCCESS;
    if (t == HSA_DEVICE_TYPE_CPU && !p->have_cpu) { p->cpu = a; p->have_cpu = true; }
    if (t == HSA_DEVICE_TYPE_GPU && !p->have_gpu) {
        uint32_t bdf = 0, domain = 0;
        const bool have_domain = hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_DOMAIN, &domain) == HSA_STATUS_SUCCESS;
        if (hsa_agent_get_info(a, (hsa_agent_info_t)HSA_AMD_AGENT_INFO_BDFID, &bdf) == HSA_STATUS_SUCCESS &&
            (p->any || (bdf == p->want_bdf && (!have_domain || domain == p->want_domain)))) {     // (8-GPU nodes repeat bus numbers across PCI domains)
            p->gpu = a; p->have_gpu = true;
        }
    }
    return HSA_STATUS_SUCCESS;
}
struct PoolPick { hsa_amd_memory_pool_t pool; bool have; };
hsa_status_t pick_pool(hsa_amd_memory_pool_t p, void* data) {
    PoolPick* pp = static_cast<PoolPick*>(data);
    hsa_amd_segment_t seg;
    bool alloc = false;
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !pp->have) { pp->pool = p; pp->have = true; }
    return HSA_STATUS_SUCCESS;
}
struct SymbolPick { const char* prefix; hsa_executable_symbol_t sym; bool have; };
hsa_status_t pick_symbol(hsa_executable_t, hsa_agent_t, hsa_executable_symbol_t s, void* data) {
    SymbolPick* sp = static_cast<SymbolPick*>(data);
    hsa_symbol_kind_t kind;
    if (hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_TYPE, &kind) != HSA_STATUS_SUCCESS || kind != HSA_SYMBOL_KIND_KERNEL) return HSA_STATUS_SUCCESS;
    uint32_t len = 0;
    hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME_LENGTH, &len);
    std::vector<char> name((size_t)len + 1, 0);
    hsa_executable_symbol_get_info(s, HSA_EXECUTABLE_SYMBOL_INFO_NAME, name.data());
    if (!sp->have && std::strncmp(name.data(), sp->prefix, std::strlen(sp->prefix)) == 0) { sp->sym = s; sp->have = true; }
    return HSA_STATUS_SUCCESS;
}
bool set_err(std::string* err, const char* what, hsa_status_t s) {
    if (err) {
        const char* m = nullptr;
        hsa_status_string(s, &m);
        *err = std::string(what) + ": " + (m ? m : "?");
    }
    return false;
}
constexpr size_t kHiddenBytes = 256;           // code object v5: the implicit arguments behind the explicit ones
constexpr hsa_signal_value_t kSignalStart = (hsa_signal_value_t)1 << 40;
// Argument buffers, used round robin: a buffer comes round again after 16 passes (four frames, > 1 GB of traffic through
// the L2s), not after two -- nothing relies on a line of an old pass having left a cache, this only makes it remote.
constexpr size_t kKernargRing = 16;

}  // namespace

void AqlQueue::on_queue_error(int /*status*/, void* /*queue*/, void* self) {
    if (self) static_cast<AqlQueue*>(self)->dead_ = true;      // the next submit() refuses, wait_idle() reports it
}

bool AqlQueue::init(int hip_device, const char* hsaco_path, const char* symbol_prefix, size_t explicit_bytes, const char* build_id, std::string* err) {
    destroy();
    dead_ = false;
    hsa_status_t st = hsa_init();              // reference-counted: HIP has it up already
    if (st != HSA_STATUS_SUCCESS) return set_err(err, "hsa_init", st);
    hsa_up_ = true;
    // the HSA agent of this HIP device: by PCI address
    AgentPick ap{};
    {
        char bus[64] = {0};
        unsigned dom = 0, b = 0, d = 0, f = 0;
        ap.any = hipDeviceGetPCIBusId(bus, sizeof bus, hip_device) != hipSuccess || std::sscanf(bus, "%x:%x:%x.%x", &dom, &b, &d, &f) < 3;
        ap.want_bdf = (b << 8) | (d << 3) | f;
        ap.want_domain = dom;
    }
    if (ap.any) { if (err) *err = "the HIP device has no PCI address to find its HSA agent by"; destroy(); return false; }
    st = hsa_iterate_agents(pick_agent, &ap);
    if (st != HSA_STATUS_SUCCESS || !ap.have_gpu || !ap.have_cpu) { if (err) *err = "no HSA agent for this HIP device"; destroy(); return false; }
    // the code object
    std::vector<char> image;
    {
        FILE* f = std::fopen(hsaco_path, "rb");
        if (!f) { if (err) *err = std::string("cannot open ") + hsaco_path; destroy(); return false; }
        std::fseek(f, 0, SEEK_END);
        const long sz = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        image.resize(sz > 0 ? (size_t)sz : 0);
        const bool ok = !image.empty() && std::fread(image.data(), 1, image.size(), f) == image.size();
        std::fclose(f);
        if (!ok) { if (err) *err = std::string("cannot read ") + hsaco_path; destroy(); return false; }
    }
    hsa_code_object_reader_t reader;
    st = hsa_code_object_reader_create_from_memory(image.data(), image.size(), &reader);
    if (st != HSA_STATUS_SUCCESS) { set_err(err, "hsa_code_object_reader_create_from_memory", st); destroy(); return false; }
    reader_ = reader.handle;
    hsa_executable_t exe;
    st = hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe);
    if (st != HSA_STATUS_SUCCESS) { set_err(err, "hsa_executable_create_alt", st); destroy(); return false; }
    executable_ = exe.handle;
    st = hsa_executable_load_agent_code_object(exe, ap.gpu, reader, nullptr, nullptr);
    if (st == HSA_STATUS_SUCCESS) st = hsa_executable_freeze(exe, nullptr);
    if (st != HSA_STATUS_SUCCESS) { set_err(err, "loading the code object", st); destroy(); return false; }
    // the code object must be THIS build's: it carries the hash of the tracker's sources and flags in a device variable
    {
        hsa_executable_symbol_t idsym;
        uint64_t addr = 0;
        uint32_t size = 0;
        char got[64] = {0};
        const size_t want_len = build_id ? std::strlen(build_id) : 0;
        st = hsa_executable_get_symbol_by_name(exe, "tsdf_track_build_id", &ap.gpu, &idsym);
        if (st == HSA_STATUS_SUCCESS) st = hsa_executable_symbol_get_info(idsym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_ADDRESS, &addr);
        if (st == HSA_STATUS_SUCCESS) st = hsa_executable_symbol_get_info(idsym, HSA_EXECUTABLE_SYMBOL_INFO_VARIABLE_SIZE, &size);
        if (st == HSA_STATUS_SUCCESS && addr && size > 0 && size < sizeof got) st = hsa_memory_copy(got, reinterpret_cast<const void*>(addr), size);
        else if (st == HSA_STATUS_SUCCESS) st = HSA_STATUS_ERROR_INVALID_SYMBOL_NAME;
        if (st != HSA_STATUS_SUCCESS || !want_len || std::strncmp(got, build_id, sizeof got) != 0) {
            if (err) *err = std::string("the code object ") + hsaco_path + " is not this library's build (id '" + got + "', wanted '" + (build_id ? build_id : "") + "')";
            destroy();
            return false;
        }
    }
    SymbolPick sp{symbol_prefix, {}, false};
    st = hsa_executable_iterate_agent_symbols(exe, ap.gpu, pick_symbol, &sp);
    if (st != HSA_STATUS_SUCCESS || !sp.have) { if (err) *err = std::string("no kernel ") + symbol_prefix + "* in the code object"; destroy(); return false; }
    hsa_executable_symbol_get_info(sp.sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kernel_object_);
    hsa_executable_symbol_get_info(sp.sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kernarg_bytes_);
    hsa_executable_symbol_get_info(sp.sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group_bytes_);
    hsa_executable_symbol_get_info(sp.sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &private_bytes_);
    const size_t explicit_aligned = (explicit_bytes + 7) & ~(size_t)7;
    if (!kernel_object_ || kernarg_bytes_ != explicit_aligned + kHiddenBytes || private_bytes_ != 0) {
        if (err) *err = "the code object's kernel does not have the argument layout this build expects (kernarg " + std::to_string(kernarg_bytes_) +
                        " bytes for " + std::to_string(explicit_bytes) + " explicit, scratch " + std::to_string(private_bytes_) + ")";
        destroy();
        return false;
    }
    explicit_bytes_ = explicit_bytes;
    // kernel arguments in device memory that the host writes through the BAR (what HIP does too: HIP_FORCE_DEV_KERNARG)
    PoolPick pp{};
    hsa_amd_agent_iterate_memory_pools(ap.gpu, pick_pool, &pp);
    kernarg_stride_ = ((size_t)kernarg_bytes_ + 4095) & ~(size_t)4095;
    if (!pp.have || hsa_amd_memory_pool_allocate(pp.pool, kKernargRing * kernarg_stride_, 0, (void**)&kernarg_) != HSA_STATUS_SUCCESS ||
        hsa_amd_agents_allow_access(1, &ap.cpu, nullptr, kernarg_) != HSA_STATUS_SUCCESS) {
        if (err) *err = "no host-writable device memory for the kernel arguments";
        destroy();
        return false;
    }
    std::memset(kernarg_, 0, kKernargRing * kernarg_stride_);
    // one signal counts DOWN over all packets (every completion subtracts one): the queue is idle when it has come down by
    // as many as were submitted -- re-arming a signal per packet would race with the completion of the packet before
    hsa_signal_t sig;
    st = hsa_signal_create(kSignalStart, 0, nullptr, &sig);
    if (st != HSA_STATUS_SUCCESS) { set_err(err, "hsa_signal_create", st); destroy(); return false; }
    signal_ = sig.handle;
    hsa_queue_t* q = nullptr;
    st = hsa_queue_create(ap.gpu, 64, HSA_QUEUE_TYPE_SINGLE,
                          [](hsa_status_t s, hsa_queue_t* qq, void* self) { AqlQueue::on_queue_error((int)s, qq, self); }, this, UINT32_MAX, UINT32_MAX, &q);
    if (st != HSA_STATUS_SUCCESS) { set_err(err, "hsa_queue_create", st); destroy(); return false; }
    queue_ = q;
    return true;
}

bool AqlQueue::submit(const void* explicit_args, uint32_t workgroups, uint32_t block) {
    hsa_queue_t* q = static_cast<hsa_queue_t*>(queue_);
    if (!q || dead_) return false;
    const uint64_t idx = hsa_queue_load_write_index_relaxed(q);
    if (idx - hsa_queue_load_read_index_scacquire(q) >= q->size) return false;       // ring full
    {   // an argument buffer comes round again after kKernargRing submissions: never while its packet may still be waiting
        hsa_signal_t sig; sig.handle = signal_;
        const uint64_t completed = (uint64_t)(kSignalStart - hsa_signal_load_relaxed(sig));
        if (submitted_ - completed >= kKernargRing) return false;
    }
    char* ka = kernarg_ + (size_t)(submitted_ % kKernargRing) * kernarg_stride_;
    std::memcpy(ka, explicit_args, explicit_bytes_);
    // code object v5 hidden arguments (tsdf_kernels.hip uses gridDim.x; blockDim is a compile-time constant there)
    char* hid = ka + ((explicit_bytes_ + 7) & ~(size_t)7);
    const uint32_t counts[3] = {workgroups, 1u, 1u};
    const uint16_t sizes[6] = {(uint16_t)block, 1, 1, 0, 0, 0};                       // group size x y z, remainders
    std::memcpy(hid + 0, counts, sizeof counts);
    std::memcpy(hid + 12, sizes, sizeof sizes);
    const uint16_t dims = 1;
    std::memcpy(hid + 64, &dims, sizeof dims);
    // The arguments must BE in device memory before the doorbell is rung: the stores above go through write-combining
    // buffers and the PCIe BAR, the doorbell takes another path inside the GPU.  As the HIP runtime does for its own
    // device-resident kernel arguments: drain the buffers, then read one byte back -- a PCIe read does not pass the posted
    // writes in front of it, so when it returns they have been performed (~0.9 us of host time per submission).
#if defined(__SSE2__)
    _mm_sfence();
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
    {
        const volatile unsigned char* last = reinterpret_cast<const volatile unsigned char*>(hid + 65);
        readback_sink_ += *last;
    }
    hsa_kernel_dispatch_packet_t* p = static_cast<hsa_kernel_dispatch_packet_t*>(q->base_address) + (idx & (q->size - 1));
    p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
    p->workgroup_size_x = (uint16_t)block; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
    p->reserved0 = 0;
    p->grid_size_x = workgroups * block; p->grid_size_y = 1; p->grid_size_z = 1;
    p->private_segment_size = private_bytes_; p->group_segment_size = group_bytes_;
    p->kernel_object = kernel_object_;
    p->kernarg_address = ka;
    p->reserved2 = 0;
    hsa_signal_t sig; sig.handle = signal_;
    p->completion_signal = sig;
    const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                            (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
    __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)p->setup << 16), __ATOMIC_RELEASE);
    hsa_queue_store_write_index_screlease(q, idx + 1);
    hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
    ++submitted_;
    return true;
}

bool AqlQueue::wait_idle() {
    if (!queue_ || !submitted_) return !dead_;
    hsa_signal_t sig; sig.handle = signal_;
    // bounded: a pass is tens of microseconds; give up after 2 s rather than hang the caller
    const hsa_signal_value_t idle = kSignalStart - (hsa_signal_value_t)submitted_;
    for (int i = 0; i < 200 && !dead_; ++i)          // 200 x 10 ms (the hint is in 100 MHz timestamp ticks)
        if (hsa_signal_wait_scacquire(sig, HSA_SIGNAL_CONDITION_LT, idle + 1, 1000000ull, HSA_WAIT_STATE_BLOCKED) <= idle) return !dead_;
    dead_ = true;                                    // a packet that never completed (or a queue fault): no further submissions
    return false;
}

void AqlQueue::destroy() {
    // A packet that never completed may still run: stop the queue from fetching, then LEAK what such a packet could touch
    // (argument ring, code object) rather than free it under a running kernel.
    const bool stuck = queue_ && !wait_idle();
    if (queue_) {
        if (stuck) hsa_queue_inactivate(static_cast<hsa_queue_t*>(queue_));
        hsa_queue_destroy(static_cast<hsa_queue_t*>(queue_)); queue_ = nullptr;
    }
    if (signal_) { hsa_signal_t s; s.handle = signal_; hsa_signal_destroy(s); signal_ = 0; }
    if (stuck) { kernarg_ = nullptr; executable_ = 0; reader_ = 0; }
    if (kernarg_) { hsa_amd_memory_pool_free(kernarg_); kernarg_ = nullptr; }
    if (executable_) { hsa_executable_t e; e.handle = executable_; hsa_executable_destroy(e); executable_ = 0; }
    if (reader_) { hsa_code_object_reader_t r; r.handle = reader_; hsa_code_object_reader_destroy(r); reader_ = 0; }
    if (hsa_up_) { hsa_shut_down(); hsa_up_ = false; }      // (drops this object's reference only)
    submitted_ = 0;
}

}  // namespace tsdf
