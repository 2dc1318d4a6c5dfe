// preproc_kernels.hip -- depth pre-processing on the GPU: the step right before the
// hot path.  The reference does it on the host with PCL (sdf_reconstruction.cpp:29-49: point-cloud conversion,
// pcl::FastBilateralFilter with its defaults, pcl::IntegralImageNormalEstimation AVERAGE_3D_GRADIENT,
// maxDepthChangeFactor 0.02, normalSmoothingSize 10).  PCL is not part of the reference tree and not in this
// image, so PARITY WITH PCL IS UNPINNED: the kernels below implement this repository's own, explicitly
// specified stand-in (tests/preproc_ref.py is its NumPy statement, the parity target):
//
//   cloud      z = depth * scale (0 -> NaN); x = (u - cx)/fx * z, y = (v - cy)/fy * z  (raw depth, as PCL's
//              filter leaves x and y alone and only replaces z)
//   bilateral  z'(p) = sum_q ws(|p-q|) wr(z(q)-z(p)) z(q) / sum_q ws wr   over the (2R+1)^2 window, valid q only,
//              ws = exp(-|p-q|^2 / (2 sigma_s^2)), wr = exp(-dz^2 / (2 sigma_r^2)); taps visited row by row
//   normals    3-D central differences of the filtered cloud (dh = P(u+1,v) - P(u-1,v), dv = P(u,v+1) - P(u,v-1),
//              rejected across depth discontinuities |dz| > 2 * max_depth_change * z), averaged over the
//              (2r+1)^2 window of valid gradients, n = normalize(dv x dh), flipped toward the camera (n.P < 0)
//
// Both windowed kernels stage their tile (+ halo) in LDS: one HBM read per input pixel.
#include <hip/hip_runtime.h>
#include <math.h>

#include "tsdf_device.h"

namespace tsdf {

constexpr int kTile = 16;                 // 16 x 16 output pixels per workgroup (256 threads)
constexpr int kMaxBilateralRadius = 32;   // LDS tile (16 + 64)^2 floats = 25.6 KB
constexpr int kMaxNormalRadius = 8;

__device__ __forceinline__ bool nan_f(float f) { return f != f; }

// ---- depth (uint16 * scale, or float metres) -> z plane
__global__ __launch_bounds__(256) void depth_to_z_kernel(const uint16_t* __restrict__ d16, const float* __restrict__ dflt,
                                                          float scale, int n, float* __restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float qnan = __int_as_float(0x7fc00000);
    float v;
    if (d16) { const uint16_t r = d16[i]; v = r ? (float)r * scale : qnan; }
    else { v = dflt[i]; if (!(v > 0.0f)) v = qnan; }
    z[i] = v;
}

// ---- bilateral filter on z
__global__ __launch_bounds__(256) void bilateral_kernel(const float* __restrict__ z, int w, int h, int R,
                                                         float inv2ss, float inv2sr, float* __restrict__ zf) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* tile = reinterpret_cast<float*>(smem_raw);
    const int T = kTile + 2 * R;
    const int x0 = blockIdx.x * kTile - R, y0 = blockIdx.y * kTile - R;
    const float qnan = __int_as_float(0x7fc00000);
    // Invalid pixels (NaN, or outside the image) sit in the tile as 1e30: dz^2 overflows to +inf, the weight is
    // exp(-inf) = 0 and 0 * 1e30 = 0, so such a tap adds +0.0f to both sums -- the same bits as skipping it (the sums
    // are never -0.0f), without a NaN test per tap.
    const float kInvalid = 1.0e30f;
    for (int t = threadIdx.x; t < T * T; t += blockDim.x) {
        const int ty = t / T, tx = t - ty * T;
        const int gx = x0 + tx, gy = y0 + ty;
        float v = kInvalid;
        if (gx >= 0 && gx < w && gy >= 0 && gy < h) { const float zz = z[gy * w + gx]; if (!nan_f(zz)) v = zz; }
        tile[t] = v;
    }
    __syncthreads();
    const int lx = threadIdx.x % kTile, ly = threadIdx.x / kTile;
    const int px = blockIdx.x * kTile + lx, py = blockIdx.y * kTile + ly;
    if (px >= w || py >= h) return;
    const float zc = tile[(ly + R) * T + lx + R];
    float out = qnan;
    if (zc < 1.0e29f) {
        float num = 0.0f, den = 0.0f;
        for (int dy = -R; dy <= R; ++dy) {
            const float* rowp = &tile[(ly + R + dy) * T + lx + R];
            for (int dx = -R; dx <= R; ++dx) {
                const float zq = rowp[dx];
                const float dz = zq - zc;
                // __expf = v_exp_f32(x * log2 e): ~1e-6 relative, 5x cheaper than ocml's expf; 3721 taps per pixel
                const float wgt = __expf(-((float)(dx * dx + dy * dy) * inv2ss) - (dz * dz) * inv2sr);
                num += wgt * zq;
                den += wgt;
            }
        }
        out = num / den;
    }
    zf[py * w + px] = out;
}

// ---- back-projection with the raw z for x,y and the filtered z
__global__ __launch_bounds__(256) void backproject_kernel(const float* __restrict__ z, const float* __restrict__ zf, int w, int h,
                                                           float fx, float fy, float cx, float cy, float* __restrict__ xyz) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const int u = i % w, v = i / w;
    const float zr = z[i], zz = zf[i];
    const float qnan = __int_as_float(0x7fc00000);
    if (nan_f(zr) || nan_f(zz)) { xyz[3 * i] = qnan; xyz[3 * i + 1] = qnan; xyz[3 * i + 2] = qnan; return; }
    xyz[3 * i + 0] = ((float)u - cx) / fx * zr;
    xyz[3 * i + 1] = ((float)v - cy) / fy * zr;
    xyz[3 * i + 2] = zz;
}

// ---- normals: windowed average of 3-D central-difference gradients
__global__ __launch_bounds__(256) void normals_kernel(const float* __restrict__ xyz, int w, int h, int r,
                                                       float max_change, float* __restrict__ nrm) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // tile of gradients: T x T entries of {dh.xyz, dv.xyz, valid}
    const int T = kTile + 2 * r;
    float* g = reinterpret_cast<float*>(smem_raw);        // 7 floats per entry, SoA planes of T*T
    const int plane = T * T;
    const int x0 = blockIdx.x * kTile - r, y0 = blockIdx.y * kTile - r;
    const float qnan = __int_as_float(0x7fc00000);
    for (int t = threadIdx.x; t < plane; t += blockDim.x) {
        const int ty = t / T, tx = t - ty * T;
        const int gx = x0 + tx, gy = y0 + ty;
        float dh[3] = {0.f, 0.f, 0.f}, dv[3] = {0.f, 0.f, 0.f}, ok = 0.0f;
        if (gx >= 1 && gx < w - 1 && gy >= 1 && gy < h - 1) {
            const float* c = &xyz[3 * (gy * w + gx)];
            const float* l = c - 3; const float* rr = c + 3;
            const float* up = c - 3 * w; const float* dn = c + 3 * w;
            const float zc = c[2];
            const bool valid = !nan_f(zc) && !nan_f(l[2]) && !nan_f(rr[2]) && !nan_f(up[2]) && !nan_f(dn[2]);
            if (valid) {
                const float lim = 2.0f * max_change * zc;
                if (fabsf(rr[2] - l[2]) <= lim && fabsf(dn[2] - up[2]) <= lim) {
                    for (int a = 0; a < 3; ++a) { dh[a] = rr[a] - l[a]; dv[a] = dn[a] - up[a]; }
                    ok = 1.0f;
                }
            }
        }
        for (int a = 0; a < 3; ++a) { g[a * plane + t] = dh[a]; g[(3 + a) * plane + t] = dv[a]; }
        g[6 * plane + t] = ok;
    }
    __syncthreads();
    const int lx = threadIdx.x % kTile, ly = threadIdx.x / kTile;
    const int px = blockIdx.x * kTile + lx, py = blockIdx.y * kTile + ly;
    if (px >= w || py >= h) return;
    const float* P = &xyz[3 * (py * w + px)];
    float n[3] = {qnan, qnan, qnan};
    if (!nan_f(P[2])) {
        float sh[3] = {0.f, 0.f, 0.f}, sv[3] = {0.f, 0.f, 0.f}, cnt = 0.0f;
        for (int dy = -r; dy <= r; ++dy)
            for (int dx = -r; dx <= r; ++dx) {
                const int t = (ly + r + dy) * T + lx + r + dx;
                const float ok = g[6 * plane + t];
                if (ok == 0.0f) continue;
                for (int a = 0; a < 3; ++a) { sh[a] += g[a * plane + t]; sv[a] += g[(3 + a) * plane + t]; }
                cnt += 1.0f;
            }
        if (cnt > 0.0f) {
            // n = dv x dh
            float c0 = sv[1] * sh[2] - sv[2] * sh[1];
            float c1 = sv[2] * sh[0] - sv[0] * sh[2];
            float c2 = sv[0] * sh[1] - sv[1] * sh[0];
            const float len = sqrtf(c0 * c0 + c1 * c1 + c2 * c2);
            if (len > 0.0f) {
                c0 /= len; c1 /= len; c2 /= len;
                if (c0 * P[0] + c1 * P[1] + c2 * P[2] > 0.0f) { c0 = -c0; c1 = -c1; c2 = -c2; }   // face the camera
                n[0] = c0; n[1] = c1; n[2] = c2;
            }
        }
    }
    float* o = &nrm[3 * (py * w + px)];
    o[0] = n[0]; o[1] = n[1]; o[2] = n[2];
}

hipError_t launch_preproc(hipStream_t s, const uint16_t* d16, const float* dflt, float scale, int w, int h,
                          const float* K /*fx fy cx cy*/, int R, float sigma_s, float sigma_r, int nr, float max_change,
                          float* z, float* zf, float* xyz, float* nrm) {
    if (R < 0 || R > kMaxBilateralRadius || nr < 1 || nr > kMaxNormalRadius) return hipErrorInvalidValue;
    const int n = w * h;
    depth_to_z_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(d16, dflt, scale, n, z);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const dim3 grid((w + kTile - 1) / kTile, (h + kTile - 1) / kTile);
    if (R > 0) {
        const int T = kTile + 2 * R;
        bilateral_kernel<<<grid, dim3(256), (size_t)T * T * sizeof(float), s>>>(z, w, h, R, 1.0f / (2.0f * sigma_s * sigma_s),
                                                                             1.0f / (2.0f * sigma_r * sigma_r), zf);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    } else {
        e = hipMemcpyAsync(zf, z, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, s);
        if (e != hipSuccess) return e;
    }
    backproject_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(z, zf, w, h, K[0], K[1], K[2], K[3], xyz);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int Tn = kTile + 2 * nr;
    normals_kernel<<<grid, dim3(256), (size_t)Tn * Tn * 7 * sizeof(float), s>>>(xyz, w, h, nr, max_change, nrm);
    return hipGetLastError();
}

}  // namespace tsdf
