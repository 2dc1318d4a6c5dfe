// h2d_probe.hip -- what does one host-to-device copy of a frame cost on this box?  (round 6)
// A 640x480 frame is 8.3 MB as planes (xyz | nrm | rgb), 1.5 MB as raw depth + rgb.  For page-locked source buffers of these
// sizes: the time of ONE hipMemcpyAsync + wait (latency form) and of 20 copies queued back to back on a stream (throughput
// form), alone and next to a kernel that keeps the chip's memory system busy (a float4 read-modify-write sweep over 2 GiB).
// Build: hipcc --offload-arch=gfx950 -O2 -o build/h2d_probe tools/h2d_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void busy(float4* p, size_t n, int reps) {
    for (int r = 0; r < reps; ++r)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; v.x += 1.f; p[i] = v; }
}
using clk = std::chrono::steady_clock;
static double us(clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); }
int main() {
    const size_t sizes[5] = {547840, 1536000, 3686400, 4608000, 8294400};   // sample list, depth16 + rgb, one plane, xyz + rgb, whole frame
    char *pin, *dev; float4* big;
    CHECK(hipHostMalloc((void**)&pin, 16 << 20, hipHostMallocDefault)); std::memset(pin, 1, 16 << 20);
    CHECK(hipMalloc((void**)&dev, 16 << 20));
    const size_t nbig = (size_t)1 << 27;                                    // 2 GiB of float4
    CHECK(hipMalloc((void**)&big, nbig * sizeof(float4)));
    hipStream_t s, k;
    CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CHECK(hipStreamCreateWithFlags(&k, hipStreamNonBlocking));
    for (int with_kernel = 0; with_kernel < 2; ++with_kernel)
        for (int si = 0; si < 5; ++si) {
            const size_t b = sizes[si];
            if (with_kernel) busy<<<2048, 256, 0, k>>>(big, nbig, 40);     // ~40 sweeps of 4 GiB traffic: tens of ms
            double one = 1e30, many = 1e30;
            for (int rep = 0; rep < 7; ++rep) {
                auto t0 = clk::now();
                CHECK(hipMemcpyAsync(dev, pin, b, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s));
                const double t = us(t0, clk::now()); if (t < one) one = t;
            }
            for (int rep = 0; rep < 5; ++rep) {
                auto t0 = clk::now();
                for (int q = 0; q < 20; ++q) CHECK(hipMemcpyAsync(dev, pin, b, hipMemcpyHostToDevice, s));
                CHECK(hipStreamSynchronize(s));
                const double t = us(t0, clk::now()) / 20.0; if (t < many) many = t;
            }
            CHECK(hipStreamSynchronize(k));
            printf("{\"bytes\": %zu, \"next_to_a_bandwidth_bound_kernel\": %s, \"one_copy_and_wait_us\": %.1f, \"per_copy_back_to_back_us\": %.1f, \"GBs_back_to_back\": %.1f}\n",
                   b, with_kernel ? "true" : "false", one, many, (double)b / many * 1e-3);
        }
    return 0;
}
