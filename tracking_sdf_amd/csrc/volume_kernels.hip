// volume_kernels.hip -- constructor fill (sdf.cpp:28-34) and the (de)interleave helpers of tsdf_download / tsdf_upload.
#include <hip/hip_runtime.h>

#include "tsdf_device.h"

namespace tsdf {

// ------------------------------------------------------------------------------------------------
// volume fill: D = width+height+depth, W = 0, Color_W = 0, R = G = B = 0.4f   (sdf.cpp:28-34)

__global__ __launch_bounds__(256) void fill_kernel(float2* __restrict__ dw, float4* __restrict__ crgb,
                                                    long long n, float d0) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        dw[i] = make_float2(d0, 0.0f);
        if (crgb) crgb[i] = make_float4(0.0f, 0.4f, 0.4f, 0.4f);
    }
}

hipError_t launch_fill(hipStream_t s, const Grid& g, float2* dw, float4* crgb, float d0) {
    const long long n = (long long)grid_stored_layers(g) * g.m * g.m;
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(dw, crgb, n, d0);
    return hipGetLastError();
}
// ------------------------------------------------------------------------------------------------
// (de)interleave helpers for tsdf_download / tsdf_upload (reference-order host mirrors)

__global__ __launch_bounds__(256) void split_kernel(const float2* __restrict__ dw, float* __restrict__ d,
                                                     float* __restrict__ w, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float2 v = dw[i];
        d[i] = v.x; w[i] = v.y;
    }
}
__global__ __launch_bounds__(256) void merge_kernel(float2* __restrict__ dw, const float* __restrict__ d,
                                                     const float* __restrict__ w, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dw[i] = make_float2(d[i], w[i]);
}
__global__ __launch_bounds__(256) void split4_kernel(const float4* __restrict__ c, float* __restrict__ a,
                                                      float* __restrict__ r, float* __restrict__ g,
                                                      float* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 v = c[i];
        a[i] = v.x; r[i] = v.y; g[i] = v.z; b[i] = v.w;
    }
}
__global__ __launch_bounds__(256) void merge4_kernel(float4* __restrict__ c, const float* __restrict__ a,
                                                      const float* __restrict__ r, const float* __restrict__ g,
                                                      const float* __restrict__ b, long long n) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        c[i] = make_float4(a[i], r[i], g[i], b[i]);
}

static inline unsigned stream_blocks(long long n) {
    long long b = (n + 255) / 256;
    if (b > 256 * 16) b = 256 * 16;
    if (b < 1) b = 1;
    return (unsigned)b;
}
hipError_t launch_split(hipStream_t s, const float2* dw, float* d, float* w, int64_t n) {
    split_kernel<<<dim3(stream_blocks(n)), dim3(256), 0, s>>>(dw, d, w, n);
    return hipGetLastError();
}
hipError_t launch_merge(hipStream_t s, float2* dw, const float* d, const float* w, int64_t n) {
    merge_kernel<<<dim3(stream_blocks(n)), dim3(256), 0, s>>>(dw, d, w, n);
    return hipGetLastError();
}
hipError_t launch_split4(hipStream_t s, const float4* c, float* a, float* r, float* g, float* b, int64_t n) {
    split4_kernel<<<dim3(stream_blocks(n)), dim3(256), 0, s>>>(c, a, r, g, b, n);
    return hipGetLastError();
}
hipError_t launch_merge4(hipStream_t s, float4* c, const float* a, const float* r, const float* g, const float* b, int64_t n) {
    merge4_kernel<<<dim3(stream_blocks(n)), dim3(256), 0, s>>>(c, a, r, g, b, n);
    return hipGetLastError();
}

}  // namespace tsdf
