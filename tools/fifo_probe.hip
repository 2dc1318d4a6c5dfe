// Does mixing L2-hit gathers with HBM streaming loads in one CU's vector-memory pipeline cost more than doing them in
// separate phases?  (tools/fifo_probe.hip, round 3: diagnosis of integrate_kernel's memory side.)
//   mixed : every wave alternates 2 scattered 16-byte gathers (2 MB table, L2 resident) and 1 coalesced 1-KiB stream load
//   phased: the same loads, but a 1024-thread workgroup (one per CU) does 8 gathers per wave, barrier, 4 stream loads, barrier
//   gather / stream: each kind alone
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 mixed, 1 phased, 2 gather only, 3 stream only
__global__ __launch_bounds__(1024) void probe(const u4* __restrict__ table, unsigned tmask, const u4* __restrict__ stream,
                                              size_t stream_elems, int iters, unsigned* __restrict__ out) {
    const unsigned tid = threadIdx.x, lane = tid & 63, wave = blockIdx.x * 16 + (tid >> 6);
    unsigned acc = 0, seed = wave * 2654435761u + lane * 40503u;
    size_t spos = ((size_t)wave * iters * 4) * 64 + lane;        // each wave streams its own contiguous region
    for (int it = 0; it < iters; ++it) {
        u4 g[8], s[4];
        if (MODE == 0) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                seed = seed * 1664525u + 1013904223u; g[2 * q] = table[(seed >> 8) & tmask];
                seed = seed * 1664525u + 1013904223u; g[2 * q + 1] = table[(seed >> 8) & tmask];
                s[q] = __builtin_nontemporal_load(&stream[(spos + (size_t)q * 64) % stream_elems]);
            }
        } else {
            if (MODE != 3) {
#pragma unroll
                for (int q = 0; q < 8; ++q) { seed = seed * 1664525u + 1013904223u; g[q] = table[(seed >> 8) & tmask]; }
            }
            if (MODE == 1) {
#pragma unroll
                for (int q = 0; q < 8; ++q) acc ^= g[q].x + g[q].w;
                __syncthreads();
            }
            if (MODE != 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] = __builtin_nontemporal_load(&stream[(spos + (size_t)q * 64) % stream_elems]);
            }
        }
        if (MODE != 3 && MODE != 1) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc ^= g[q].x + g[q].w;
        }
        if (MODE != 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc ^= s[q].y + s[q].z;
        }
        if (MODE == 1) __syncthreads();
        spos += 4 * 64;
    }
    if (acc == 0x12345678u) out[wave] = acc;
}

int main() {
    const size_t tbytes = 2u << 20, sbytes = (size_t)2 << 30;
    u4 *table, *stream; unsigned* out;
    CHECK(hipMalloc(&table, tbytes)); CHECK(hipMalloc(&stream, sbytes)); CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemset(table, 1, tbytes)); CHECK(hipMemset(stream, 2, sbytes));
    const int blocks = 256, iters = 64;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const char* names[4] = {"mixed", "phased", "gather_only", "stream_only"};
    for (int rep = 0; rep < 3; ++rep)
        for (int mode = 0; mode < 4; ++mode) {
            CHECK(hipEventRecord(a));
            for (int k = 0; k < 5; ++k) {
                const unsigned tm = (unsigned)(tbytes / 16 - 1); const size_t se = sbytes / 16;
                if (mode == 0) probe<0><<<blocks, 1024>>>(table, tm, stream, se, iters, out);
                if (mode == 1) probe<1><<<blocks, 1024>>>(table, tm, stream, se, iters, out);
                if (mode == 2) probe<2><<<blocks, 1024>>>(table, tm, stream, se, iters, out);
                if (mode == 3) probe<3><<<blocks, 1024>>>(table, tm, stream, se, iters, out);
            }
            CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            const double streamed = (double)blocks * 16 * iters * 4 * 1024, gathered = (double)blocks * 16 * iters * 8 * 1024;
            if (rep == 2) printf("{\"mode\": \"%s\", \"us_per_launch\": %.1f, \"stream_GBs\": %.0f, \"gather_GBs\": %.0f}\n", names[mode], ms * 200.0,
                                 mode == 2 ? 0.0 : streamed / (ms / 5 * 1e-3) / 1e9, mode == 3 ? 0.0 : gathered / (ms / 5 * 1e-3) / 1e9);
        }
    return 0;
}
