// launch_probe.hip -- host-side cost of submitting one kernel, by submission path and by the size of its arguments.
//   paths (1.3 KB of arguments): the <<<>>> syntax (hipLaunchKernel with an argument-pointer array, what the library
//   uses), hipModuleLaunchKernel with the arguments pre-marshalled in one buffer, hipExtLaunchKernel;
//   sizes: <<<>>> with 64 B .. 3840 B of arguments.
// Build: hipcc --offload-arch=gfx950 -O2 -o build/launch_probe tools/launch_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstring>
template <int N> struct Args { double v[N]; };
template <int N> __global__ void k(Args<N> b, double* out) { if (threadIdx.x == 0 && blockIdx.x == 0 && b.v[3] == 12345.0) out[0] = b.v[N - 1]; }
static auto now() { return std::chrono::steady_clock::now(); }
template <class A, class B> static double us(A a, B c) { return std::chrono::duration<double, std::micro>(c - a).count(); }
template <int N> static double chevron(hipStream_t s, double* out, int n) {
    Args<N> b; std::memset(&b, 0, sizeof b);
    double t = 0;
    for (int i = 0; i < n; ++i) {
        auto t0 = now();
        k<N><<<dim3(714), dim3(384), 0, s>>>(b, out);
        t += us(t0, now());
        if ((i & 63) == 63) (void)hipStreamSynchronize(s);   // keep the queue shallow: the call never blocks on a full ring
    }
    return t / n;
}
int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    double* out; (void)hipMalloc(&out, 8);
    const int n = 20000;
    for (int rep = 0; rep < 2; ++rep) {
        const double a = chevron<160>(s, out, n);
        double tb = 0, tc = 0;
        Args<160> b; std::memset(&b, 0, sizeof b);
        hipFunction_t f;
        if (hipGetFuncBySymbol(&f, (const void*)k<160>) == hipSuccess) {
            struct { Args<160> b; double* out; } args; args.b = b; args.out = out;
            size_t sz = sizeof args;
            void* extra[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, &args, HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
            for (int i = 0; i < n; ++i) {
                auto t0 = now();
                (void)hipModuleLaunchKernel(f, 714, 1, 1, 384, 1, 1, 0, s, nullptr, extra);
                tb += us(t0, now());
                if ((i & 63) == 63) (void)hipStreamSynchronize(s);
            }
        }
        void* argp[] = {&b, &out};
        for (int i = 0; i < n; ++i) {
            auto t0 = now();
            (void)hipExtLaunchKernel((const void*)k<160>, dim3(714), dim3(384), argp, 0, s, nullptr, nullptr, 0);
            tc += us(t0, now());
            if ((i & 63) == 63) (void)hipStreamSynchronize(s);
        }
        std::printf("{\"launch_call_us_by_path_1280B\": {\"triple_chevron\": %.3f, \"module_launch_prepacked\": %.3f, \"ext_launch\": %.3f}, "
                    "\"launch_call_us_by_argument_bytes\": {\"64\": %.3f, \"512\": %.3f, \"1280\": %.3f, \"2560\": %.3f, \"3840\": %.3f}}\n",
                    a, tb / n, tc / n, chevron<8>(s, out, n), chevron<64>(s, out, n), chevron<160>(s, out, n), chevron<320>(s, out, n), chevron<480>(s, out, n));
    }
    return 0;
}
