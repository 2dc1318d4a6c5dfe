// Cost of integrate_kernel's pixel gather by record format (round 3 diagnosis; volume traffic as in tools/mix_probe.hip).
// FORMAT 0: 32-byte records, paired 16-byte pieces, 2 instructions   (the kernel with colour)
//        1: 24-byte records, paired 12-byte pieces, 2 instructions   (the kernel without colour)
//        2: 16-byte records, one 16-byte load per lane, 1 instruction
//        3:  8-byte records, one  8-byte load per lane, 1 instruction
//        4:  4-byte records, one  4-byte load per lane, 1 instruction
// ACTIVE: lanes that gather (a contiguous run of that many lanes, the others carry a dropped offset).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef unsigned u4 __attribute__((ext_vector_type(4)));
typedef unsigned u3 __attribute__((ext_vector_type(3)));
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s >> 8; }

template <int FORMAT, int VOLUME>
__global__ __launch_bounds__(256) void probe(const char* __restrict__ rec, unsigned nrec, u2* __restrict__ dw, u4* __restrict__ col,
                                             unsigned nseg, int items_per_wave, unsigned active, unsigned* __restrict__ out) {
    const unsigned lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
    unsigned seed = wave * 2654435761u + 12345u, acc = 0;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(rec), 0, (int)(nrec * 32u), 0x00020000);
    for (int it = 0; it < items_per_wave; ++it) {
        const unsigned base = rnd(seed) % nrec;
        const unsigned seg = __builtin_amdgcn_readfirstlane(rnd(seed) % nseg);
        const unsigned first = __builtin_amdgcn_readfirstlane(rnd(seed) % 30u);
        const unsigned gfirst = __builtin_amdgcn_readfirstlane(rnd(seed) % (65u - active));
        auto pix = [&](unsigned l) { return (base + l + (l >> 1)) % nrec; };               // pixels 1..2 apart
        auto on = [&](unsigned l) { return l >= gfirst && l < gfirst + active; };
        if (FORMAT == 0) {
            const unsigned la = lane >> 1, lb = 32 + (lane >> 1);
            const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, on(la) ? (int)(pix(la) * 32 + (lane & 1) * 16) : 0x7fffffff, 0, 0);
            const u4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, on(lb) ? (int)(pix(lb) * 32 + (lane & 1) * 16) : 0x7fffffff, 0, 0);
            acc ^= a.x + b.w;
        } else if (FORMAT == 1) {
            const unsigned la = lane >> 1, lb = 32 + (lane >> 1);
            const u3 a = __builtin_amdgcn_raw_buffer_load_b96(rs, on(la) ? (int)(pix(la) * 24 + (lane & 1) * 12) : 0x7fffffff, 0, 0);
            const u3 b = __builtin_amdgcn_raw_buffer_load_b96(rs, on(lb) ? (int)(pix(lb) * 24 + (lane & 1) * 12) : 0x7fffffff, 0, 0);
            acc ^= a.x + b.z;
        } else if (FORMAT == 2) {
            const u4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, on(lane) ? (int)(pix(lane) * 16) : 0x7fffffff, 0, 0);
            acc ^= a.x + a.w;
        } else if (FORMAT == 3) {
            const u2 a = __builtin_amdgcn_raw_buffer_load_b64(rs, on(lane) ? (int)(pix(lane) * 8) : 0x7fffffff, 0, 0);
            acc ^= a.x + a.y;
        } else {
            acc ^= __builtin_amdgcn_raw_buffer_load_b32(rs, on(lane) ? (int)(pix(lane) * 4) : 0x7fffffff, 0, 0);
        }
        if (VOLUME) {
            const bool live = lane >= first && lane < first + 34u;
            u2 d = u2{0, 0}; u4 c = u4{0, 0, 0, 0};
            if (live) { d = dw[(size_t)seg * 64 + lane]; c = __builtin_nontemporal_load(&col[(size_t)seg * 64 + lane]); }
            d.x += acc; c.y ^= d.y;
            if (live) { dw[(size_t)seg * 64 + lane] = d; __builtin_nontemporal_store(c, &col[(size_t)seg * 64 + lane]); }
        }
    }
    if (acc == 0x12345678u) out[wave] = acc;
}

int main() {
    const unsigned nrec = 307200, nseg = 2097152;
    char* rec; u4* col; u2* dw; unsigned* out;
    CHECK(hipMalloc(&rec, (size_t)nrec * 32)); CHECK(hipMalloc(&dw, (size_t)nseg * 512)); CHECK(hipMalloc(&col, (size_t)nseg * 1024));
    CHECK(hipMalloc(&out, 1 << 20));
    CHECK(hipMemset(rec, 1, (size_t)nrec * 32)); CHECK(hipMemset(dw, 0, (size_t)nseg * 512)); CHECK(hipMemset(col, 0, (size_t)nseg * 1024));
    const int blocks = 1280, ipw = 39;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const unsigned actives[3] = {64, 51, 34};
    for (int vol = 0; vol < 2; ++vol)
        for (int fmt = 0; fmt < 5; ++fmt)
            for (int ai = 0; ai < 3; ++ai) {
                float best = 1e9f;
                for (int rep = 0; rep < 3; ++rep) {
                    CHECK(hipEventRecord(a));
                    for (int k = 0; k < 10; ++k) {
#define L(F, V) probe<F, V><<<blocks, 256>>>(rec, nrec, dw, col, nseg, ipw, actives[ai], out)
                        if (vol == 0) { if (fmt == 0) L(0, 0); if (fmt == 1) L(1, 0); if (fmt == 2) L(2, 0); if (fmt == 3) L(3, 0); if (fmt == 4) L(4, 0); }
                        else { if (fmt == 0) L(0, 1); if (fmt == 1) L(1, 1); if (fmt == 2) L(2, 1); if (fmt == 3) L(3, 1); if (fmt == 4) L(4, 1); }
                    }
                    CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
                    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
                    if (ms < best) best = ms;
                }
                printf("{\"with_volume_traffic\": %d, \"format\": %d, \"gathering_lanes\": %u, \"us_per_launch\": %.1f}\n", vol, fmt, actives[ai], best * 100.0);
            }
    return 0;
}
