// rmw_probe.hip -- the measured HBM ceiling the roofline is quoted against: plain streaming
// read-modify-write kernels over arrays shaped like the volume, timed with HIP events.
//
//   dw      float2 {D,W} per voxel                 16 B of traffic per voxel   (integrate without colour)
//   dw+rgb  float2 + float4 {Color_W,R,G,B}        48 B per voxel              (integrate with colour)
//   read    float2, read only                      8 B per voxel               (mesh count pass)
//   rows    dw+rgb, but in the integrate kernel's granularity: one wavefront takes one 64-voxel item
//           (512 B + 1 KiB) at a time, items of a row consecutive, every other row of the volume skipped
//
// Build + run on the GPU box:  make rmw_probe && build/rmw_probe [voxels_per_axis=512]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void rmw_dw(float2* __restrict__ dw, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float2 v = dw[i];
        v.y += 1.0f; v.x = (v.x * 0.5f + 0.25f) / v.y;
        dw[i] = v;
    }
}

__global__ __launch_bounds__(256) void rmw_dw_rgb(float2* __restrict__ dw, float4* __restrict__ c, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float2 v = dw[i];
        float4 q = c[i];
        v.y += 1.0f; v.x = (v.x * 0.5f + 0.25f) / v.y;
        q.x += 0.5f; q.y = (q.y + 1.0f) / q.x; q.z = (q.z + 2.0f) / q.x; q.w = (q.w + 3.0f) / q.x;
        dw[i] = v;
        c[i] = q;
    }
}

__global__ __launch_bounds__(256) void read_dw(const float2* __restrict__ dw, long long n, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    float acc = 0.0f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 v = dw[i];
        acc += v.x * v.y;
    }
    if (acc == 123.456f) out[0] = acc;
}

// persistent grid, one 64-voxel item per wavefront step, contiguous item ranges per wavefront
__global__ __launch_bounds__(256) void rmw_rows(float2* __restrict__ dw, float4* __restrict__ c, int m, long long n_rows_used) {
    const int lane = threadIdx.x & 63;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const int chunks = m / 64;
    const long long n_items = n_rows_used * chunks;
    const long long per = (n_items + n_waves - 1) / n_waves;
    const long long first = wave * per, last = first + per < n_items ? first + per : n_items;
    for (long long it = first; it < last; ++it) {
        const long long row = 2 * (it / chunks);                 // every other row: strided like a clipped frustum
        const long long i = row * m + (it % chunks) * 64 + lane;
        float2 v = dw[i];
        float4 q = c[i];
        v.y += 1.0f; v.x = (v.x * 0.5f + 0.25f) / v.y;
        q.x += 0.5f; q.y = (q.y + 1.0f) / q.x; q.z = (q.z + 2.0f) / q.x; q.w = (q.w + 3.0f) / q.x;
        dw[i] = v;
        c[i] = q;
    }
}

int main(int argc, char** argv) {
    const int m = argc > 1 ? std::atoi(argv[1]) : 512;
    const long long n = (long long)m * m * m;
    float2* dw; float4* c; float* out;
    CHECK(hipMalloc(&dw, n * sizeof(float2)));
    CHECK(hipMalloc(&c, n * sizeof(float4)));
    CHECK(hipMalloc(&out, 64));
    CHECK(hipMemset(dw, 0, n * sizeof(float2)));
    CHECK(hipMemset(c, 0, n * sizeof(float4)));
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    const int grid = 256 * 8, reps = 10;
    const char* names[4] = {"dw", "dw+rgb", "read", "rows"};
    const double bytes[4] = {16.0 * n, 48.0 * n, 8.0 * n, 48.0 * (n / 2)};
    std::printf("{\"m\": %d", m);
    for (int k = 0; k < 4; ++k) {
        float best = 1e30f;
        for (int r = 0; r < reps + 2; ++r) {
            CHECK(hipEventRecord(a));
            if (k == 0) rmw_dw<<<grid, 256>>>(dw, n);
            else if (k == 1) rmw_dw_rgb<<<grid, 256>>>(dw, c, n);
            else if (k == 2) read_dw<<<grid, 256>>>(dw, n, out);
            else rmw_rows<<<256 * 4, 256>>>(dw, c, m, ((long long)m * m) / 2);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0.f;
            CHECK(hipEventElapsedTime(&ms, a, b));
            if (r >= 2 && ms < best) best = ms;
        }
        std::printf(", \"%s_ms\": %.4f, \"%s_GBs\": %.1f", names[k], best, names[k], bytes[k] / (best * 1e-3) / 1e9);
    }
    std::printf("}\n");
    return 0;
}
