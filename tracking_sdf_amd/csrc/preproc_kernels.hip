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
//              (grid_filter = 0), or, grid_filter = 1 (default), the bilateral GRID of Paris & Durand 2006 -- the
//              algorithm family pcl::FastBilateralFilter belongs to -- in O(pixels + cells) instead of
//              O(pixels * window):  cells of sigma_s x sigma_s pixels x sigma_r metres, 2 cells of padding;
//                splat   every valid pixel adds (z, 1) to its NEAREST cell; the z sum of a cell is exact (integers
//                        trunc(z * 2^32)) and rounded to float once, so the order of the pixels does not matter;
//                blur    per axis (x, y, z in turn) two passes of [1 2 1]/4 over the cells that are interior in all
//                        three axes; cells on the faces of the grid keep their (zero) splat value;
//                slice   z'(p) = S/W of the trilinear interpolation of (S, W) at (u/sigma_s+2, v/sigma_s+2,
//                        (z-zmin)/sigma_r+2), the eight corners added in x-fastest order.
//              Every float operation of the grid path is a single correctly rounded f32 operation in a fixed order,
//              so it is BIT-EXACT against tests/preproc_ref.py (the windowed path differs in expf's last bits).
//   normals    3-D central differences of the filtered cloud (dh = P(u+1,v) - P(u-1,v), dv = P(u,v+1) - P(u,v-1),
//              rejected across depth discontinuities |dz| > 2 * max_depth_change * z), averaged over the
//              (2r+1)^2 window of valid gradients (box sums done separably: along the rows left to right, then
//              the row sums top to bottom), n = normalize(dv x dh), flipped toward the camera (n.P < 0)
//
// Both windowed kernels stage their tile (+ halo) in LDS: one HBM read per input pixel.
#include <hip/hip_runtime.h>
#include <math.h>

#include "tsdf_device.h"

namespace tsdf {

constexpr int kTile = 16;                 // 16 x 16 output pixels per workgroup (256 threads)
constexpr int kMaxBilateralRadius = 32;   // LDS tile (16 + 64)^2 floats = 25.6 KB
constexpr int kMaxNormalRadius = 8;
constexpr int kGridPad = 2;               // empty cells around the bilateral grid, every axis
constexpr int kSplatMaxDepthCells = 4096; // depth cells of one (x, y) cell column: 12 B of LDS each

__device__ __forceinline__ bool nan_f(float f) { return f != f; }

// A depth is a measurement when it is a positive number below kMaxDepthMetres: zero / negative / NaN mean "no
// reading", and so do +inf ("too far" in ROS float images, REP-117) and absurd ranges.
constexpr float kMaxDepthMetres = 1000.0f;
__device__ __forceinline__ bool depth_ok(float v) { return v > 0.0f && v < kMaxDepthMetres; }

// ---- depth (uint16 * scale, or float metres) -> z plane
__global__ __launch_bounds__(256) void depth_to_z_kernel(const uint16_t* __restrict__ d16, const float* __restrict__ dflt,
                                                          float scale, int n, float* __restrict__ z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float qnan = __int_as_float(0x7fc00000);
    float v;
    if (d16) { const uint16_t r = d16[i]; v = (float)r * scale; }
    else v = dflt[i];
    z[i] = depth_ok(v) ? v : qnan;
}

// ---- the same, plus min / max of the valid depths (positive floats order like their bit patterns):
// mm[0] = min bits, mm[1] = ~(max bits), both initialised to 0xffffffff by the caller and lowered with atomicMin
__global__ __launch_bounds__(256) void depth_to_z_minmax_kernel(const uint16_t* __restrict__ d16, const float* __restrict__ dflt,
                                                                 float scale, int n, float* __restrict__ z, unsigned* __restrict__ mm) {
    __shared__ unsigned s_lo[4], s_hi[4];
    const float qnan = __int_as_float(0x7fc00000);
    unsigned lo = 0xffffffffu, hic = 0xffffffffu;
    const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 8;           // 8 consecutive pixels per thread
    float v[8];
    if (i0 + 8 <= n) {
        if (d16) {
            const uint4 q = *reinterpret_cast<const uint4*>(d16 + i0);
            const unsigned r[4] = {q.x, q.y, q.z, q.w};
            for (int k = 0; k < 4; ++k) {
                const unsigned a = r[k] & 0xffffu, b = r[k] >> 16;
                v[2 * k] = (float)a * scale; v[2 * k + 1] = (float)b * scale;
            }
            for (int k = 0; k < 8; ++k) if (!depth_ok(v[k])) v[k] = qnan;
        } else {
            const float4 a = *reinterpret_cast<const float4*>(dflt + i0), b = *reinterpret_cast<const float4*>(dflt + i0 + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
            for (int k = 0; k < 8; ++k) if (!depth_ok(v[k])) v[k] = qnan;
        }
        *reinterpret_cast<float4*>(z + i0) = make_float4(v[0], v[1], v[2], v[3]);
        *reinterpret_cast<float4*>(z + i0 + 4) = make_float4(v[4], v[5], v[6], v[7]);
    } else {
        for (int k = 0; k < 8; ++k) {
            v[k] = qnan;
            if (i0 + k < n) {
                const float f = d16 ? (float)d16[i0 + k] * scale : dflt[i0 + k];
                if (depth_ok(f)) v[k] = f;
                z[i0 + k] = v[k];
            }
        }
    }
    for (int k = 0; k < 8; ++k)
        if (v[k] > 0.0f) { const unsigned b = __float_as_uint(v[k]); lo = min(lo, b); hic = min(hic, ~b); }
    for (int o = 32; o > 0; o >>= 1) { lo = min(lo, (unsigned)__shfl_xor((int)lo, o)); hic = min(hic, (unsigned)__shfl_xor((int)hic, o)); }
    if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6] = lo; s_hi[threadIdx.x >> 6] = hic; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) { lo = min(lo, s_lo[k]); hic = min(hic, s_hi[k]); }
        if (lo != 0xffffffffu) { atomicMin(&mm[0], lo); atomicMin(&mm[1], hic); }
    }
}

// ---- bilateral grid, splat: one workgroup per (cx, cy) cell column, one thread per pixel of its block (at most
// (floor(sigma_s)+1)^2 pixels).  The depth sum of a cell is EXACT: every z enters as the integer trunc(z * 2^32)
// (0.23 nm units; exact for z >= 2 mm), the integers are added with LDS atomics -- an exact sum has no order -- and
// the total is rounded to float once.  z < 1024 m and <= 1024 pixels per cell keep the total below 2^52, so the
// int64 -> double -> float conversion rounds once.  Every cell of the column is written (zeros included): no memset.
__device__ __forceinline__ int grid_cell_xy(int x, float sigma_s) { return (int)((float)x / sigma_s + 0.5f); }

__device__ void splat_block_range(int b, int extent, float sigma_s, int* lo_out, int* hi_out) {
    int lo = (int)floorf(((float)b - 0.5f) * sigma_s) - 1;
    if (lo < 0) lo = 0;
    if (lo > extent) lo = extent;
    while (lo < extent && grid_cell_xy(lo, sigma_s) < b) ++lo;
    int hi = (int)floorf(((float)b + 0.5f) * sigma_s) - 1;    // at most the first pixel of the next cell
    if (hi < lo) hi = lo;
    if (hi > extent) hi = extent;
    while (hi < extent && grid_cell_xy(hi, sigma_s) <= b) ++hi;
    *lo_out = lo; *hi_out = hi;
}

__global__ __launch_bounds__(256) void grid_splat_kernel(const float* __restrict__ z, int w, int h, float sigma_s, float sigma_r,
                                                          float zmin, int gy, int gz, float2* __restrict__ grid) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned long long* s_sum = reinterpret_cast<unsigned long long*>(smem_raw);          // gz exact sums
    unsigned* s_cnt = reinterpret_cast<unsigned*>(smem_raw + (size_t)gz * sizeof(unsigned long long));
    __shared__ int s_rng[4];
    const int cx = blockIdx.x, cy = blockIdx.y;
    if (threadIdx.x == 0) splat_block_range(cx - kGridPad, w, sigma_s, &s_rng[0], &s_rng[1]);
    if (threadIdx.x == 64) splat_block_range(cy - kGridPad, h, sigma_s, &s_rng[2], &s_rng[3]);
    for (int cz = threadIdx.x; cz < gz; cz += blockDim.x) { s_sum[cz] = 0ull; s_cnt[cz] = 0u; }
    __syncthreads();
    const int x0 = s_rng[0], bw = s_rng[1] - s_rng[0], y0 = s_rng[2], bh = s_rng[3] - s_rng[2];
    const int np = bw * bh;
    for (int t = threadIdx.x; t < np; t += blockDim.x) {
        const int ty = t / bw, tx = t - ty * bw;
        const float v = z[(y0 + ty) * w + x0 + tx];
        if (!nan_f(v)) {
            const int cz = (int)((v - zmin) / sigma_r + 0.5f) + kGridPad;
            atomicAdd(&s_sum[cz], (unsigned long long)((double)v * 4294967296.0));
            atomicAdd(&s_cnt[cz], 1u);
        }
    }
    __syncthreads();
    float2* col = grid + ((size_t)cx * gy + cy) * gz;
    for (int cz = threadIdx.x; cz < gz; cz += blockDim.x)
        col[cz] = make_float2((float)((double)(long long)s_sum[cz] * (1.0 / 4294967296.0)), (float)s_cnt[cz]);
}

// ---- bilateral grid, blur along one axis: BOTH [1 2 1]/4 passes in one launch (the three first-pass values a
// second-pass cell needs are recomputed, same operations and roundings as two ping-pong passes).  `off` = element
// stride of the axis, `len` its extent, `c` the cell's coordinate along it.
__device__ __forceinline__ float2 blur_tap(const float2* __restrict__ p, int off) {
    const float2 a = p[-off], b = p[off], c = p[0];
    return make_float2((a.x + b.x + 2.0f * c.x) * 0.25f, (a.y + b.y + 2.0f * c.y) * 0.25f);
}
__global__ __launch_bounds__(256) void grid_blur_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                         int gx, int gy, int gz, int axis) {
    const int n = gx * gy * gz;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int cz = i % gz, cy = (i / gz) % gy, cx = i / (gz * gy);
    const bool face = cx == 0 || cx == gx - 1 || cy == 0 || cy == gy - 1 || cz == 0 || cz == gz - 1;
    float2 r = in[i];
    if (!face) {
        const int off = axis == 0 ? gy * gz : axis == 1 ? gz : 1;
        const int len = axis == 0 ? gx : axis == 1 ? gy : gz;
        const int c = axis == 0 ? cx : axis == 1 ? cy : cz;
        const float2 m0 = blur_tap(in + i, off);                                         // first pass at c (interior)
        const float2 ml = c - 1 == 0 ? in[i - off] : blur_tap(in + i - off, off);        // ... at c-1 (a face cell stays)
        const float2 mr = c + 1 == len - 1 ? in[i + off] : blur_tap(in + i + off, off);  // ... at c+1
        r = make_float2((ml.x + mr.x + 2.0f * m0.x) * 0.25f, (ml.y + mr.y + 2.0f * m0.y) * 0.25f);
    }
    out[i] = r;
}

// ---- bilateral grid, slice + back-projection: z' = S/W of the trilinear interpolation; x, y from the raw depth
__global__ __launch_bounds__(256) void grid_slice_backproject_kernel(const float* __restrict__ z, const float2* __restrict__ grid,
                                                                      int w, int h, float sigma_s, float sigma_r, float zmin,
                                                                      int gx, int gy, int gz, float fx, float fy, float cx, float cy,
                                                                      float* __restrict__ zf, float* __restrict__ xyz) {
#pragma clang fp contract(off)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= w * h) return;
    const int u = i % w, v = i / w;
    const float zr = z[i];
    const float qnan = __int_as_float(0x7fc00000);
    if (nan_f(zr)) { zf[i] = qnan; xyz[3 * i] = qnan; xyz[3 * i + 1] = qnan; xyz[3 * i + 2] = qnan; return; }
    const float px = (float)u / sigma_s + (float)kGridPad, py = (float)v / sigma_s + (float)kGridPad;
    const float pz = (zr - zmin) / sigma_r + (float)kGridPad;
    int xi = (int)px, yi = (int)py, zi = (int)pz;
    const float xa = px - (float)xi, ya = py - (float)yi, za = pz - (float)zi;
    xi = min(max(xi, 0), gx - 1); yi = min(max(yi, 0), gy - 1); zi = min(max(zi, 0), gz - 1);
    const int xx = min(xi + 1, gx - 1), yy = min(yi + 1, gy - 1), zz = min(zi + 1, gz - 1);
    const float xb = 1.0f - xa, yb = 1.0f - ya, zb = 1.0f - za;
    const size_t sx = (size_t)gy * gz;
    const float2 c000 = grid[xi * sx + (size_t)yi * gz + zi], c100 = grid[xx * sx + (size_t)yi * gz + zi];
    const float2 c010 = grid[xi * sx + (size_t)yy * gz + zi], c110 = grid[xx * sx + (size_t)yy * gz + zi];
    const float2 c001 = grid[xi * sx + (size_t)yi * gz + zz], c101 = grid[xx * sx + (size_t)yi * gz + zz];
    const float2 c011 = grid[xi * sx + (size_t)yy * gz + zz], c111 = grid[xx * sx + (size_t)yy * gz + zz];
    const float w000 = xb * yb * zb, w100 = xa * yb * zb, w010 = xb * ya * zb, w110 = xa * ya * zb;
    const float w001 = xb * yb * za, w101 = xa * yb * za, w011 = xb * ya * za, w111 = xa * ya * za;
    float S = w000 * c000.x, W = w000 * c000.y;
    S = S + w100 * c100.x; W = W + w100 * c100.y;
    S = S + w010 * c010.x; W = W + w010 * c010.y;
    S = S + w110 * c110.x; W = W + w110 * c110.y;
    S = S + w001 * c001.x; W = W + w001 * c001.y;
    S = S + w101 * c101.x; W = W + w101 * c101.y;
    S = S + w011 * c011.x; W = W + w011 * c011.y;
    S = S + w111 * c111.x; W = W + w111 * c111.y;
    const float out = S / W;
    zf[i] = out;
    if (nan_f(out)) { xyz[3 * i] = qnan; xyz[3 * i + 1] = qnan; xyz[3 * i + 2] = qnan; return; }
    xyz[3 * i + 0] = ((float)u - cx) / fx * zr;
    xyz[3 * i + 1] = ((float)v - cy) / fy * zr;
    xyz[3 * i + 2] = out;
}

// ---- bilateral filter on z, windowed
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
    // box sums, separably (the integral-image idea of the PCL estimator): row sums left to right over the 2r+1
    // columns for all T rows of the tile, then those added top to bottom -- 2(2r+1) adds per value instead of (2r+1)^2
    float* rs = g + 7 * plane;                               // 7 planes of T x kTile row sums
    const int rplane = T * kTile;
    for (int t = threadIdx.x; t < rplane; t += blockDim.x) {
        const int ty = t / kTile, tx = t - ty * kTile;
        float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        const int base = ty * T + tx;
        for (int dx = 0; dx <= 2 * r; ++dx)
            for (int a = 0; a < 7; ++a) acc[a] += g[a * plane + base + dx];
        for (int a = 0; a < 7; ++a) rs[a * rplane + t] = acc[a];
    }
    __syncthreads();
    const int lx = threadIdx.x % kTile, ly = threadIdx.x / kTile;
    const int px = blockIdx.x * kTile + lx, py = blockIdx.y * kTile + ly;
    if (px >= w || py >= h) return;
    const float* P = &xyz[3 * (py * w + px)];
    float n[3] = {qnan, qnan, qnan};
    if (!nan_f(P[2])) {
        float sh[3] = {0.f, 0.f, 0.f}, sv[3] = {0.f, 0.f, 0.f}, cnt = 0.0f;
        for (int dy = 0; dy <= 2 * r; ++dy) {
            const int t = (ly + dy) * kTile + lx;
            for (int a = 0; a < 3; ++a) { sh[a] += rs[a * rplane + t]; sv[a] += rs[(3 + a) * rplane + t]; }
            cnt += rs[6 * rplane + t];
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

hipError_t launch_depth_to_z(hipStream_t s, const uint16_t* d16, const float* dflt, float scale, int n, float* z,
                             unsigned* minmax) {
    if (minmax) {
        hipError_t e = hipMemsetAsync(minmax, 0xff, 2 * sizeof(unsigned), s);
        if (e != hipSuccess) return e;
        depth_to_z_minmax_kernel<<<dim3((n + 2047) / 2048), dim3(256), 0, s>>>(d16, dflt, scale, n, z, minmax);
    } else {
        depth_to_z_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(d16, dflt, scale, n, z);
    }
    return hipGetLastError();
}

// pixels at or beyond zcut become "no reading" (a frame whose depth range needs more cells than one cell column holds)
__global__ __launch_bounds__(256) void clip_far_kernel(float* __restrict__ z, int n, float zcut) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && z[i] >= zcut) z[i] = __int_as_float(0x7fc00000);
}

bool bilateral_grid_plan(int w, int h, float sigma_s, float sigma_r, float zmin, float zmax, BilateralGrid* g) {
    if (!(sigma_s >= 1.0f) || !(sigma_s <= 30.0f) || !(sigma_r > 0.0f) || !(zmax >= zmin) || !(zmax < 1024.0f)) return false;
    float nz = (zmax - zmin) / sigma_r;
    const float nz_limit = (float)(kSplatMaxDepthCells - 1 - 2 * kGridPad);
    g->cut = 0; g->zcut = zmax;
    if (!(nz < nz_limit)) {
        // the range does not fit the depth cells of a cell column: keep the near part, drop the pixels behind it
        // (one outlier at 300 m must not cost the frame) -- same rule in tests/preproc_ref.py
        g->cut = 1;
        g->zcut = zmin + sigma_r * (nz_limit - 1.0f);
        nz = (g->zcut - zmin) / sigma_r;
        if (!(nz < nz_limit)) return false;
    }
    g->gx = (int)((float)(w - 1) / sigma_s) + 1 + 2 * kGridPad;
    g->gy = (int)((float)(h - 1) / sigma_s) + 1 + 2 * kGridPad;
    g->gz = (int)nz + 1 + 2 * kGridPad;
    g->zmin = zmin;
    return true;
}

hipError_t launch_preproc(hipStream_t s, int w, int h, const float* K /*fx fy cx cy*/, int R, float sigma_s, float sigma_r,
                          int nr, float max_change, const BilateralGrid* bg, float2* grid_a, float2* grid_b,
                          const float* z, float* zf, float* xyz, float* nrm) {
    if (R < 0 || R > kMaxBilateralRadius || nr < 1 || nr > kMaxNormalRadius) return hipErrorInvalidValue;
    const int n = w * h;
    hipError_t e;
    const dim3 grid((w + kTile - 1) / kTile, (h + kTile - 1) / kTile);
    if (bg && R > 0) {
        if (bg->cut) clip_far_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(const_cast<float*>(z), n, bg->zcut);
        const int cells = bg->gx * bg->gy * bg->gz;
        grid_splat_kernel<<<dim3(bg->gx, bg->gy), dim3(256), (size_t)bg->gz * 12, s>>>(z, w, h, sigma_s, sigma_r, bg->zmin, bg->gy, bg->gz, grid_a);
        grid_blur_kernel<<<dim3((cells + 255) / 256), dim3(256), 0, s>>>(grid_a, grid_b, bg->gx, bg->gy, bg->gz, 0);
        grid_blur_kernel<<<dim3((cells + 255) / 256), dim3(256), 0, s>>>(grid_b, grid_a, bg->gx, bg->gy, bg->gz, 1);
        grid_blur_kernel<<<dim3((cells + 255) / 256), dim3(256), 0, s>>>(grid_a, grid_b, bg->gx, bg->gy, bg->gz, 2);
        grid_slice_backproject_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(z, grid_b, w, h, sigma_s, sigma_r, bg->zmin,
                                                                                 bg->gx, bg->gy, bg->gz, K[0], K[1], K[2], K[3], zf, xyz);
        e = hipGetLastError();
        if (e != hipSuccess) return e;
    } else {
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
    }
    const int Tn = kTile + 2 * nr;
    normals_kernel<<<grid, dim3(256), (size_t)(Tn * Tn + Tn * kTile) * 7 * sizeof(float), s>>>(xyz, w, h, nr, max_change, nrm);
    return hipGetLastError();
}

}  // namespace tsdf
