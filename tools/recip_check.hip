// recip_check.hip -- exhaustive check of recip_ieee_small (tsdf_kernels.hip) against the compiler's IEEE division:
// every float in [1e-5f, 4.0f].  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -o build/recip_check tools/recip_check.hip && build/recip_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>

__device__ __forceinline__ float recip_ieee_small(float v) {
    float r = __builtin_amdgcn_rcpf(v);
    const float e0 = __builtin_fmaf(-v, r, 1.0f);
    r = __builtin_fmaf(e0, r, r);
    float q = r;
    const float e1 = __builtin_fmaf(-v, q, 1.0f);
    q = __builtin_fmaf(e1, r, q);
    const float e2 = __builtin_fmaf(-v, q, 1.0f);
    return __builtin_fmaf(e2, r, q);
}

__global__ void check(unsigned lo, unsigned hi, unsigned long long* bad, unsigned* first_bad) {
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long b = lo + (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; b <= hi; b += stride) {
        const float v = __uint_as_float((unsigned)b);
        const float want = 1.0f / v;
        const float got = recip_ieee_small(v);
        if (__float_as_uint(want) != __float_as_uint(got)) {
            if (atomicAdd(bad, 1ull) == 0ull) *first_bad = (unsigned)b;
        }
    }
}

int main() {
    float lo_f = 1.0e-5f, hi_f = 4.0f;
    unsigned lo, hi;
    std::memcpy(&lo, &lo_f, 4); std::memcpy(&hi, &hi_f, 4);
    unsigned long long* bad; unsigned* first;
    hipMalloc(&bad, 8); hipMalloc(&first, 4); hipMemset(bad, 0, 8); hipMemset(first, 0, 4);
    check<<<4096, 256>>>(lo, hi, bad, first);
    unsigned long long n = 0; unsigned f = 0;
    hipMemcpy(&n, bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, first, 4, hipMemcpyDeviceToHost);
    std::printf("{\"values\": %llu, \"mismatches\": %llu, \"first_mismatch_bits\": \"0x%08x\"}\n", (unsigned long long)hi - lo + 1, n, f);
    return n ? 1 : 0;
}
