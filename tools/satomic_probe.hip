// satomic_probe.hip -- do gfx950's SCALAR returning atomics work, and what does a shared cursor cost?
// Every wavefront of a persistent grid (CUs x 5 workgroups x 4 wavefronts, as integrate_kernel's) draws tickets from one of
// P cursors with s_atomic_add (one returning atomic per WAVEFRONT, scalar unit, lgkmcnt) until the cursor passes `items`;
// a ticket marks its entry in a table.  Checks: every entry marked exactly once.  Reports the kernel time per draw.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/satomic_probe tools/satomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned draw(unsigned* cursor, unsigned n) {
    unsigned v = n;
    const unsigned long long a = (unsigned long long)cursor;
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
    const unsigned long long ua = ((unsigned long long)hi << 32) | lo;        // wave-uniform: lives in an SGPR pair
    asm volatile("s_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(v) : "s"(ua) : "memory");
    return v;
}

__global__ __launch_bounds__(256) void probe(unsigned* cursors, unsigned pools, unsigned items_per_pool, unsigned* marks, int work) {
    const unsigned wave = (blockIdx.x * 256u + threadIdx.x) >> 6;
    const unsigned pool = wave % pools;
    unsigned* cur = cursors + 32u * pool;                 // one 128-byte line per cursor
    float acc = 0.f;
    for (unsigned guard = 0; guard < (1u << 20); ++guard) {     // bounded: never spins forever
        const unsigned t = draw(cur, 1u);
        if (t >= items_per_pool) break;
        if ((threadIdx.x & 63u) == 0) atomicAdd(&marks[pool * items_per_pool + t], 1u);
        for (int k = 0; k < work; ++k) acc = acc * 1.0001f + (float)k;      // stand-in for an item's arithmetic
    }
    if (acc == 12345.678f) marks[0] = 7;
}

int main() {
    int cus = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    const int blocks = cus * 5;
    const unsigned total = 200000;
    for (unsigned pools : {8u, 32u, 128u, 512u}) {
        for (int work : {0, 400}) {
            const unsigned per = total / pools;
            unsigned *cursors, *marks;
            hipMalloc(&cursors, pools * 128);
            hipMalloc(&marks, (size_t)pools * per * 4);
            hipMemset(cursors, 0, pools * 128);
            hipMemset(marks, 0, (size_t)pools * per * 4);
            hipEvent_t a, b;
            hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            probe<<<blocks, 256>>>(cursors, pools, per, marks, work);
            hipEventRecord(b);
            if (hipDeviceSynchronize() != hipSuccess) { std::printf("kernel failed\n"); return 1; }
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            std::vector<unsigned> m((size_t)pools * per);
            hipMemcpy(m.data(), marks, m.size() * 4, hipMemcpyDeviceToHost);
            size_t bad = 0;
            for (unsigned v : m) bad += v != 1u;
            std::printf("{\"pools\": %u, \"items\": %u, \"work_iterations_per_item\": %d, \"kernel_us\": %.1f, \"ns_per_draw_chipwide\": %.2f, \"entries_not_marked_exactly_once\": %zu}\n",
                        pools, pools * per, work, ms * 1e3, ms * 1e6 / (pools * per), bad);
            hipFree(cursors); hipFree(marks);
        }
    }
    return 0;
}
