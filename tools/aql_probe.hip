// aql_probe.hip -- is a hand-written AQL dispatch cheaper than hipLaunchKernel for a tracker pass?
//
// A Gauss-Newton pass is one launch of 714 x 384 threads whose last workgroup hands a row to the host; the host solves and
// launches the next pass.  Round 4 measured 2.8 us inside the launch CALL and ~8 us per pass between the kernel's 15.4 us and
// the pass's 23.5 us of wall time.  This probe submits the same SHAPE of work -- 714 x 384 threads, every workgroup arrives
// on a counter, the last one stores a word into pinned host memory that the host spins on -- in two ways:
//   hip   the <<<>>> launch on a HIP stream, arguments passed by value (what the library does)
//   aql   a user-mode HSA queue of this process's own: the dispatch packet is built once, the host patches 16 bytes of
//         kernel arguments (they live in DEVICE memory, written through the PCIe BAR), stores the packet header and
//         rings the doorbell -- no runtime call on the path
// and reports, per submission: host time inside the submission, and the round trip submission -> word seen by the host.
// Build (make aql_probe):
//   hipcc --offload-arch=gfx950 -O2 --genco --no-gpu-bundle-output -o build/aql_probe_kernel.hsaco tools/aql_probe.hip -DPROBE_DEVICE_ONLY
//   hipcc --offload-arch=gfx950 -O2 -o build/aql_probe tools/aql_probe.hip -lhsa-runtime64
#include <hip/hip_runtime.h>

// the kernel: no blockDim / gridDim (code-object-v5 kernels read those from hidden arguments the probe does not fill)
struct ProbeArgs { unsigned* counter; unsigned long long* host_word; unsigned long long seq; unsigned n_wg; unsigned pad; };
extern "C" __global__ __launch_bounds__(1024) void probe_kernel(ProbeArgs a) {
    __shared__ int last;
    // a little of what a tracker workgroup does before it arrives: a dependent load and a barrier
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        last = old == a.n_wg - 1u;
    }
    __syncthreads();
    if (last && threadIdx.x == 0) {
        __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(a.host_word, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

#ifndef PROBE_DEVICE_ONLY
#include <hsa/hsa.h>
#include <hsa/hsa_ext_amd.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define HSA_OK(x) do { hsa_status_t s__ = (x); if (s__ != HSA_STATUS_SUCCESS) { const char* m = nullptr; hsa_status_string(s__, &m); \
    std::fprintf(stderr, "%s failed: %s\n", #x, m ? m : "?"); return 1; } } while (0)
#define HIP_OK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { std::fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

static hsa_agent_t g_gpu, g_cpu;
static bool g_have_gpu = false, g_have_cpu = false;
static hsa_status_t pick_agent(hsa_agent_t a, void*) {
    hsa_device_type_t t;
    hsa_agent_get_info(a, HSA_AGENT_INFO_DEVICE, &t);
    if (t == HSA_DEVICE_TYPE_GPU && !g_have_gpu) { g_gpu = a; g_have_gpu = true; }
    if (t == HSA_DEVICE_TYPE_CPU && !g_have_cpu) { g_cpu = a; g_have_cpu = true; }
    return HSA_STATUS_SUCCESS;
}
static hsa_amd_memory_pool_t g_dev_pool;
static bool g_have_dev_pool = false;
static hsa_status_t pick_pool(hsa_amd_memory_pool_t p, void*) {
    hsa_amd_segment_t seg;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_SEGMENT, &seg);
    bool alloc = false;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_RUNTIME_ALLOC_ALLOWED, &alloc);
    uint32_t flags = 0;
    hsa_amd_memory_pool_get_info(p, HSA_AMD_MEMORY_POOL_INFO_GLOBAL_FLAGS, &flags);
    if (seg == HSA_AMD_SEGMENT_GLOBAL && alloc && (flags & HSA_AMD_MEMORY_POOL_GLOBAL_FLAG_COARSE_GRAINED) && !g_have_dev_pool) { g_dev_pool = p; g_have_dev_pool = true; }
    return HSA_STATUS_SUCCESS;
}

using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
static double median(std::vector<double> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main(int argc, char** argv) {
    const char* hsaco = argc > 1 ? argv[1] : "build/aql_probe_kernel.hsaco";
    const int n = argc > 2 ? std::atoi(argv[2]) : 20000;
    // round 6: the shape and where the word goes are arguments (tools/aql_probe_sweep.sh): what part of the 12.3 us floor of a
    // pass-shaped launch is the grid (workgroups x threads), what part the command processor, what part the PCIe store?
    const unsigned n_wg = argc > 3 ? (unsigned)std::atoi(argv[3]) : 714u, block = argc > 4 ? (unsigned)std::atoi(argv[4]) : 384u;
    const bool word_in_bar = argc > 5 && std::strcmp(argv[5], "bar") == 0;   // the word in DEVICE memory, the host polls it through the BAR
    const bool aql_only = word_in_bar;
    HIP_OK(hipSetDevice(0));
    hipStream_t stream;
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    unsigned* counter;
    HIP_OK(hipMalloc((void**)&counter, 256));
    HIP_OK(hipMemset(counter, 0, 256));
    unsigned long long* word;
    HIP_OK(hipHostMalloc((void**)&word, 64, hipHostMallocDefault));
    *word = 0;
    unsigned long long seq = 0;
    auto wait_word = [&](unsigned long long want) {
        const auto t0 = clk::now();
        for (unsigned spins = 0;; ++spins) {
            if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) return true;
            if ((spins & 4095u) == 4095u && us(t0, clk::now()) > 2e6) return false;
        }
    };

    // ---- hip: <<<>>> on a stream
    std::vector<double> hip_call, hip_trip;
    for (int i = 0; i < n + 200 && !aql_only; ++i) {
        ProbeArgs a{counter, word, ++seq, n_wg, 0};
        const auto t0 = clk::now();
        probe_kernel<<<dim3(n_wg), dim3(block), 0, stream>>>(a);
        const auto t1 = clk::now();
        if (!wait_word(seq)) { std::fprintf(stderr, "hip path: word never arrived\n"); return 1; }
        const auto t2 = clk::now();
        if (i >= 200) { hip_call.push_back(us(t0, t1)); hip_trip.push_back(us(t0, t2)); }
    }
    HIP_OK(hipStreamSynchronize(stream));

    // ---- aql: this process's own user-mode queue
    HSA_OK(hsa_init());
    HSA_OK(hsa_iterate_agents(pick_agent, nullptr));
    if (!g_have_gpu || !g_have_cpu) { std::fprintf(stderr, "no GPU / CPU agent\n"); return 1; }
    HSA_OK(hsa_amd_agent_iterate_memory_pools(g_gpu, pick_pool, nullptr));
    if (!g_have_dev_pool) { std::fprintf(stderr, "no device memory pool\n"); return 1; }
    hsa_queue_t* q = nullptr;
    HSA_OK(hsa_queue_create(g_gpu, 256, HSA_QUEUE_TYPE_SINGLE, nullptr, nullptr, UINT32_MAX, UINT32_MAX, &q));
    // the code object
    std::vector<char> image;
    {
        FILE* f = std::fopen(hsaco, "rb");
        if (!f) { std::perror(hsaco); return 1; }
        std::fseek(f, 0, SEEK_END); const long sz = std::ftell(f); std::fseek(f, 0, SEEK_SET);
        image.resize((size_t)sz);
        if (std::fread(image.data(), 1, image.size(), f) != image.size()) return 1;
        std::fclose(f);
    }
    hsa_code_object_reader_t reader;
    HSA_OK(hsa_code_object_reader_create_from_memory(image.data(), image.size(), &reader));
    hsa_executable_t exe;
    HSA_OK(hsa_executable_create_alt(HSA_PROFILE_FULL, HSA_DEFAULT_FLOAT_ROUNDING_MODE_DEFAULT, nullptr, &exe));
    HSA_OK(hsa_executable_load_agent_code_object(exe, g_gpu, reader, nullptr, nullptr));
    HSA_OK(hsa_executable_freeze(exe, nullptr));
    hsa_executable_symbol_t sym;
    HSA_OK(hsa_executable_get_symbol_by_name(exe, "probe_kernel.kd", &g_gpu, &sym));
    uint64_t kernel_object = 0;
    uint32_t kernarg_size = 0, group_size = 0, private_size = 0;
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_OBJECT, &kernel_object));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_KERNARG_SEGMENT_SIZE, &kernarg_size));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_GROUP_SEGMENT_SIZE, &group_size));
    HSA_OK(hsa_executable_symbol_get_info(sym, HSA_EXECUTABLE_SYMBOL_INFO_KERNEL_PRIVATE_SEGMENT_SIZE, &private_size));
    // kernel arguments in DEVICE memory the host can write (through the BAR), two buffers used alternately
    char* karg = nullptr;
    HSA_OK(hsa_amd_memory_pool_allocate(g_dev_pool, 8192, 0, (void**)&karg));
    HSA_OK(hsa_amd_agents_allow_access(1, &g_cpu, nullptr, karg));
    std::memset(karg, 0, 8192);
    if (word_in_bar) { word = reinterpret_cast<unsigned long long*>(karg + 2048); *word = 0; }
    const uint32_t mask = q->size - 1;
    hsa_kernel_dispatch_packet_t* ring = static_cast<hsa_kernel_dispatch_packet_t*>(q->base_address);
    std::vector<double> aql_call, aql_trip;
    for (int i = 0; i < n + 200; ++i) {
        ++seq;
        const auto t0 = clk::now();
        char* ka = karg + (size_t)(i & 1) * 4096;
        ProbeArgs a{counter, word, seq, n_wg, 0};
        std::memcpy(ka, &a, sizeof a);                         // (a real pass would patch the pose: ~0.8 KB)
        const uint64_t idx = hsa_queue_add_write_index_relaxed(q, 1);
        hsa_kernel_dispatch_packet_t* p = &ring[idx & mask];
        p->setup = 1 << HSA_KERNEL_DISPATCH_PACKET_SETUP_DIMENSIONS;
        p->workgroup_size_x = (uint16_t)block; p->workgroup_size_y = 1; p->workgroup_size_z = 1;
        p->grid_size_x = n_wg * block; p->grid_size_y = 1; p->grid_size_z = 1;
        p->private_segment_size = private_size; p->group_segment_size = group_size;
        p->kernel_object = kernel_object;
        p->kernarg_address = ka;
        p->completion_signal.handle = 0;
        const uint16_t header = (HSA_PACKET_TYPE_KERNEL_DISPATCH << HSA_PACKET_HEADER_TYPE) | (1 << HSA_PACKET_HEADER_BARRIER) |
                                (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCACQUIRE_FENCE_SCOPE) |
                                (HSA_FENCE_SCOPE_AGENT << HSA_PACKET_HEADER_SCRELEASE_FENCE_SCOPE);
        __atomic_store_n(reinterpret_cast<uint32_t*>(p), (uint32_t)header | ((uint32_t)p->setup << 16), __ATOMIC_RELEASE);
        hsa_signal_store_screlease(q->doorbell_signal, (hsa_signal_value_t)idx);
        const auto t1 = clk::now();
        if (!wait_word(seq)) { std::fprintf(stderr, "aql path: word never arrived (i = %d)\n", i); return 1; }
        const auto t2 = clk::now();
        if (i >= 200) { aql_call.push_back(us(t0, t1)); aql_trip.push_back(us(t0, t2)); }
    }
    // ---- alternating, the way the library would mix them: pass 0 on the HIP stream, passes 1.. on the queue
    if (hip_call.empty()) { hip_call.push_back(0.0); hip_trip.push_back(0.0); }
    std::printf("{\"submissions\": %d, \"workgroups\": %u, \"threads_per_workgroup\": %u, \"word\": \"%s\", "
                "\"hip_launch\": {\"call_us_median\": %.2f, \"round_trip_us_median\": %.2f}, "
                "\"aql_own_queue\": {\"submit_us_median\": %.2f, \"round_trip_us_median\": %.2f}, \"kernarg_segment_bytes\": %u}\n",
                n, n_wg, block, word_in_bar ? "device memory, host polls through the BAR" : "pinned host memory",
                median(hip_call), median(hip_trip), median(aql_call), median(aql_trip), kernarg_size);
    hsa_queue_destroy(q);
    return 0;
}
#endif
