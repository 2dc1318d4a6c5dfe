// tsdf_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the tracking_sdf hot path.
//
//   integrate_kernel   SDF::update                          (reference src/sdf.cpp:224-315)
//   track_kernel       one Gauss-Newton accumulation pass    (reference src/camera_tracking.cpp:146-189,
//                      + get_partial_derivative :246-363, SDF::interpolate_distance sdf.cpp:127-163)
//   track_final_kernel fixed-order sum of the per-workgroup partial normal equations
//   pack_kernel        per-frame image packing (xyz|nrm|rgb planes -> 32-byte pixel records + the
//                      tracker's column-major stride-3 sample list, camera_tracking.cpp:162-163)
//
// Numerics: every operation that decides a result (f64 geometry, f32 interpolation and running
// averages, (int) truncations) is the reference's operation in the reference's order, so this file
// MUST be compiled with -ffp-contract=off (no FMA contraction) and without fast-math.
// Both kernels are HBM/L2-bound byte movers: no MFMA anywhere (there is no dense contraction).
#include <hip/hip_runtime.h>
#include <limits.h>
#include <math.h>

#include "tsdf_device.h"

namespace tsdf {

// ------------------------------------------------------------------------------------------------
// small device helpers

// (int)float / (int)double as x86-64 cvttss2si / cvttsd2si: out-of-range and NaN give INT_MIN
// (the reference relies on this at sdf.cpp:143-145 and :251-252; v_cvt_i32_f32 would saturate).
__device__ __forceinline__ int trunc_x86(float f) {
    return (f >= -2147483648.0f && f < 2147483648.0f) ? (int)f : INT_MIN;
}

// Eigen 3.2 coefficient-based 3x3 * 3: ((a0*b0 + a1*b1) + a2*b2), see DESIGN.md "evaluation orders".
__device__ __forceinline__ double row3(const double* M, double x, double y, double z) {
    return (M[0] * x + M[1] * y) + M[2] * z;
}

__device__ __forceinline__ bool is_nan(float f) { return f != f; }

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
    const long long n = (long long)(g.xe - g.xs) * g.m * g.m;
    if (n <= 0) return hipSuccess;
    long long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    fill_kernel<<<dim3((unsigned)blocks), dim3(256), 0, s>>>(dw, crgb, n, d0);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// frame packing.  Pixel record = 2 x float4: {Px,Py,Pz, rgb bits} {Nx,Ny,Nz, 0}.  One 32-byte
// sector per projected voxel instead of three scattered plane reads.  Records are stored row-major
// or column-major (pix_su / pix_sv), whichever makes the pixels hit by 64 consecutive k of one voxel
// row neighbours in memory: a k-row projects to a near-vertical image line for an upright camera,
// and with row-major records every lane of the gather then pulls its own 128-byte line through L2
// (measured: 0.6 ms of L2->L1 line traffic per 512^3 frame, the v1 bottleneck).  The tracker's sample list is
// written in the reference's visiting order: columns outer, rows inner, both with `stride`.

__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ xyz, const float* __restrict__ nrm,
                                                    const uint8_t* __restrict__ rgb, int width, int height,
                                                    int stride, int pix_su, int pix_sv, float4* __restrict__ pn,
                                                    float4* __restrict__ samples, int ncols, int nrows) {
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= width * height) return;
    const float qnan = __int_as_float(0x7fc00000);
    const float px = xyz[3 * pix + 0], py = xyz[3 * pix + 1], pz = xyz[3 * pix + 2];
    float nx = qnan, ny = qnan, nz = qnan;
    if (nrm) { nx = nrm[3 * pix + 0]; ny = nrm[3 * pix + 1]; nz = nrm[3 * pix + 2]; }
    unsigned c = 0;
    if (rgb) c = (unsigned)rgb[3 * pix + 0] | ((unsigned)rgb[3 * pix + 1] << 8) | ((unsigned)rgb[3 * pix + 2] << 16);
    const int col = pix % width, row = pix / width;
    const long long rec = (long long)col * pix_su + (long long)row * pix_sv;   // row- or column-major records
    pn[2 * rec + 0] = make_float4(px, py, pz, __uint_as_float(c));
    pn[2 * rec + 1] = make_float4(nx, ny, nz, 0.0f);
    if (col % stride == 0 && row % stride == 0) {
        const int ci = col / stride, rj = row / stride;
        if (ci < ncols && rj < nrows) samples[ci * nrows + rj] = make_float4(px, py, pz, 0.0f);
    }
}

hipError_t launch_pack(hipStream_t s, const float* xyz, const float* nrm, const uint8_t* rgb,
                       int32_t width, int32_t height, int32_t stride, int32_t pix_su, int32_t pix_sv,
                       float4* pn, float4* samples, int32_t ncols, int32_t nrows) {
    const int n = width * height;
    pack_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(xyz, nrm, rgb, width, height, stride, pix_su, pix_sv, pn,
                                                              samples, ncols, nrows);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// TSDF integration.
//
// The reference visits all m^3 voxels and rejects most of them (behind the camera / outside the
// image / NaN pixel / d > delta).  Rejected voxels cost no HBM traffic, so a kernel that evaluates
// the full f64 projection for every voxel is ALU-bound on the rejects (round-1 v1: 0.61 ms at 512^3
// for 9 M updated voxels).  v2 culls per k-ROW first:
//
//   * a row (fixed i,j; k = 0..m-1) is a straight segment in camera space, pc(k) = Q0 + k Q1, and
//     each frustum test (z >= 0, u > -1, u < W, v > -1, v < H) is affine in k, so the set of k that
//     can pass is one interval [klo, khi], obtained from five divisions per ROW instead of two per
//     VOXEL.  The interval is widened by one voxel each side and every voxel inside it still runs
//     the reference's exact test, so the cull never changes a result;
//   * one workgroup = 64 consecutive rows (wave w takes rows 4r+w: 16 rows per wavefront, lanes
//     0..15 do the row clips); the wave then walks only the 64-voxel chunks that intersect
//     [klo, khi]: 64 lanes = 64 consecutive k = one 512-byte {D,W} segment, perfectly coalesced;
//   * one workgroup per tile, not persistent: tiles differ by 50x in work, the hardware dispatcher
//     is the load balancer.
//
// Algorithmic traffic: 16 B (48 B with colour) per *updated* voxel + the 32-byte pixel records.

constexpr int kRowsPerTile = 64;
constexpr int kRowsPerWave = kRowsPerTile / (kIntegrateBlock / 64);   // 16

struct IntegrateTiling {
    long long n_rows;    // (xe-xs) * m
    int log2m;           // >= 0 when m is a power of two
    int clip;            // 1 = K has the usual last row (0,0,k22>0): row clipping is valid
};

// interval of k (real-valued) on which a + k*b > 0, intersected into [lo, hi]
__device__ __forceinline__ void clip_affine(double a, double b, double& lo, double& hi) {
    if (b > 0.0) { const double t = -a / b; if (t > lo) lo = t; }
    else if (b < 0.0) { const double t = -a / b; if (t < hi) hi = t; }
    else if (a < -1.0e-9) { lo = 1.0; hi = 0.0; }   // row parallel to this plane and clearly outside it
    // (b == 0 and a within rounding of 0: leave it to the exact per-voxel test)
}

template <bool COLOR>
__global__ __launch_bounds__(kIntegrateBlock) void integrate_kernel(
    IntegrateParams p, IntegrateTiling tl, float2* __restrict__ dw, float4* __restrict__ crgb,
    const float4* __restrict__ pn, unsigned long long* __restrict__ counters) {
    const int m = p.g.m;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const double ox = p.g.origin[0], oy = p.g.origin[1], oz = p.g.origin[2];
    const double cw = (double)p.g.cell_w, ch = (double)p.g.cell_h, cd = (double)p.g.cell_d;
    const float delta = p.g.delta, eps = p.g.epsilon;
    const long long row0 = (long long)blockIdx.x * kRowsPerTile;
    unsigned n_own = 0, n_halo = 0;

    // ---- per-row clip: lane r (< 16) owns row row0 + 4 r + wv
    int klo_v = 1, khi_v = 0;
    {
        const long long row = row0 + 4 * lane + wv;
        if (lane < kRowsPerWave && row < tl.n_rows) {
            int il, j;
            if (tl.log2m >= 0) { il = (int)(row >> tl.log2m); j = (int)(row & (m - 1)); }
            else { il = (int)(row / m); j = (int)(row - (long long)il * m); }
            klo_v = 0; khi_v = m - 1;
            if (tl.clip) {
                const double gx = cw * ((double)(il + p.g.xs) + 0.5) + ox;
                const double gy = ch * ((double)j + 0.5) + oy;
                const double gz0 = cd * 0.5 + oz;               // k = 0
                // pc(k) = Q0 + k Q1
                double Q0[3], Q1[3];
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    Q0[a] = (p.rot_inv[3 * a] * gx + p.rot_inv[3 * a + 1] * gy) + p.rot_inv[3 * a + 2] * gz0 + p.rot_inv_trans[a];
                    Q1[a] = p.rot_inv[3 * a + 2] * cd;
                }
                // ij = K pc, with K's last row (0,0,k22): ij2 = k22 * pcz has the sign of pcz
                const double a0 = row3(&p.K[0], Q0[0], Q0[1], Q0[2]), b0 = row3(&p.K[0], Q1[0], Q1[1], Q1[2]);
                const double a1 = row3(&p.K[3], Q0[0], Q0[1], Q0[2]), b1 = row3(&p.K[3], Q1[0], Q1[1], Q1[2]);
                const double a2 = p.K[8] * Q0[2], b2 = p.K[8] * Q1[2];
                double lo = -1.0, hi = (double)m;
                clip_affine(a2, b2, lo, hi);                                   // pcz > 0 (>= handled by the margin)
                clip_affine(a0 + a2, b0 + b2, lo, hi);                         // u > -1
                clip_affine((double)p.width * a2 - a0, (double)p.width * b2 - b0, lo, hi);     // u < W
                clip_affine(a1 + a2, b1 + b2, lo, hi);                         // v > -1
                clip_affine((double)p.height * a2 - a1, (double)p.height * b2 - b1, lo, hi);   // v < H
                if (!(lo <= hi + 1.0e-6)) { klo_v = 1; khi_v = 0; }            // empty (or NaN): nothing can pass
                else {
                    // widen by one voxel per side; the exact test below decides inside
                    const double l2 = floor(lo) - 1.0, h2 = ceil(hi) + 1.0;
                    klo_v = l2 < 0.0 ? 0 : (l2 > (double)(m - 1) ? m : (int)l2);
                    khi_v = h2 > (double)(m - 1) ? m - 1 : (h2 < 0.0 ? -1 : (int)h2);
                }
            }
        }
    }

    // ---- walk the rows of this wave
    for (int r = 0; r < kRowsPerWave; ++r) {
        const int klo = __builtin_amdgcn_readlane(klo_v, r);
        const int khi = __builtin_amdgcn_readlane(khi_v, r);
        if (klo > khi) continue;                                   // wave-uniform
        const long long row = row0 + 4 * r + wv;
        int il, j;
        if (tl.log2m >= 0) { il = (int)(row >> tl.log2m); j = (int)(row & (m - 1)); }
        else { il = (int)(row / m); j = (int)(row - (long long)il * m); }
        const int i = il + p.g.xs;
        const bool owned = (i >= p.g.own_x0 && i < p.g.own_x1);
        // get_global_coordinates, sdf.h:153-157: (extent/(float)m) * (i + 0.5) + origin
        const double gx = cw * ((double)i + 0.5) + ox;
        const double gy = ch * ((double)j + 0.5) + oy;
        // first two terms of rot_inv * g (Eigen order: (r0*gx + r1*gy) + r2*gz), shared by the row
        const double sx = p.rot_inv[0] * gx + p.rot_inv[1] * gy;
        const double sy = p.rot_inv[3] * gx + p.rot_inv[4] * gy;
        const double sz = p.rot_inv[6] * gx + p.rot_inv[7] * gy;
        const long long row_base = row * m;

        for (int k = (klo & ~63) + lane; k <= khi; k += 64) {
            if (k < klo) continue;
            const double gz = cd * ((double)k + 0.5) + oz;
            // project_world_to_camera, camera_tracking.cpp:51-54
            const double pcx = (sx + p.rot_inv[2] * gz) + p.rot_inv_trans[0];
            const double pcy = (sy + p.rot_inv[5] * gz) + p.rot_inv_trans[1];
            const double pcz = (sz + p.rot_inv[8] * gz) + p.rot_inv_trans[2];
            if (pcz < 0) continue;                                              // sdf.cpp:247-249
            // project_camera_to_image_plane, camera_tracking.cpp:40-47
            const double ij0 = row3(&p.K[0], pcx, pcy, pcz);
            const double ij1 = row3(&p.K[3], pcx, pcy, pcz);
            const double ij2 = row3(&p.K[6], pcx, pcy, pcz);
            const double u = ij0 / ij2;
            const double w = ij1 / ij2;
            // (int) truncation toward zero + unsigned compare (sdf.cpp:251-256): pixel c is hit by
            // u in (c-1, c+1) for c = 0 and [c, c+1) otherwise; NaN / inf / overflow are rejected.
            if (!(u > -1.0 && u < (double)p.width && w > -1.0 && w < (double)p.height)) continue;
            const int iu = (int)u, iw = (int)w;
            const long long pix = (long long)iu * p.pix_su + (long long)iw * p.pix_sv;
            const float4 P = pn[2 * pix + 0];
            const float4 N = pn[2 * pix + 1];
            if (is_nan(P.x) || is_nan(P.y) || is_nan(N.x) || is_nan(N.y) || is_nan(N.z)) continue;  // :260
            // projectivePointToPlaneDistance, sdf.h:177-181 (Eigen dot: a0*b0 + (a1*b1 + a2*b2))
            const double dx = (double)P.x - pcx, dy = (double)P.y - pcy, dz = (double)P.z - pcz;
            const double nx = (double)N.x, ny = (double)N.y, nz = (double)N.z;
            const double p2p = dx * nx + (dy * ny + dz * nz);
            float d_new = (float)p2p;                                           // sdf.cpp:274
            float w_new = 1.0f;
            if (d_new >= eps && d_new <= delta) {                               // sdf.cpp:277-279
                const float a = d_new - eps;
                w_new = (float)exp((-0.5 * (double)a) * (double)a);
            }
            if (d_new > delta) continue;                                        // sdf.cpp:280-283
            if (d_new < -delta) d_new = -delta;                                 // sdf.cpp:285-287

            const long long idx = row_base + k;
            const float2 old = dw[idx];                                         // {D, W}
            const float w_sum = old.y + w_new;                                  // sdf.cpp:289-292
            const float d_out = (old.y * old.x + w_new * d_new) / w_sum;
            dw[idx] = make_float2(d_out, w_sum);
            if (owned) ++n_own; else ++n_halo;

            if (COLOR) {                                                        // sdf.cpp:294-304
                const double cosine = fabs(0.0 * nx + (0.0 * ny + 1.0 * nz)) / sqrt(nx * nx + (ny * ny + nz * nz));
                const float wc = (float)((double)w_new * cosine);
                const float4 c = crgb[idx];                                     // {Color_W, R, G, B}
                const unsigned bits = __float_as_uint(P.w);
                const float pr = (float)(int)(bits & 255u), pg = (float)(int)((bits >> 8) & 255u),
                            pb = (float)(int)((bits >> 16) & 255u);
                const float cw_sum = c.x + wc;
                crgb[idx] = make_float4(cw_sum, (c.x * c.y + wc * pr) / cw_sum, (c.x * c.z + wc * pg) / cw_sum,
                                        (c.x * c.w + wc * pb) / cw_sum);
            }
        }
    }

    // one atomic per counter per workgroup that updated anything: wave shuffle, then LDS across the 4 waves
    __shared__ unsigned s_cnt[2][kIntegrateBlock / 64];
    for (int off = 32; off > 0; off >>= 1) {
        n_own += __shfl_xor(n_own, off);
        n_halo += __shfl_xor(n_halo, off);
    }
    if (lane == 0) { s_cnt[0][wv] = n_own; s_cnt[1][wv] = n_halo; }
    __syncthreads();
    if (tid == 0) {
        unsigned a = 0, b = 0;
        for (int q = 0; q < kIntegrateBlock / 64; ++q) { a += s_cnt[0][q]; b += s_cnt[1][q]; }
        if (a) atomicAdd(&counters[kCntUpdatedOwned], (unsigned long long)a);
        if (b) atomicAdd(&counters[kCntUpdatedHalo], (unsigned long long)b);
    }
}

hipError_t launch_integrate(hipStream_t s, const IntegrateParams& p, float2* dw, float4* crgb,
                            const float4* pn, unsigned long long* counters) {
    const int m = p.g.m;
    const int nx = p.g.xe - p.g.xs;
    if (nx <= 0 || m <= 0) return hipSuccess;
    IntegrateTiling tl;
    tl.n_rows = (long long)nx * m;
    tl.log2m = -1;
    for (int b = 0; b < 31; ++b) if ((1 << b) == m) tl.log2m = b;
    tl.clip = (p.K[6] == 0.0 && p.K[7] == 0.0 && p.K[8] > 0.0) ? 1 : 0;
    const long long blocks = (tl.n_rows + kRowsPerTile - 1) / kRowsPerTile;
    if (p.with_color)
        integrate_kernel<true><<<dim3((unsigned)blocks), dim3(kIntegrateBlock), 0, s>>>(p, tl, dw, crgb, pn, counters);
    else
        integrate_kernel<false><<<dim3((unsigned)blocks), dim3(kIntegrateBlock), 0, s>>>(p, tl, dw, crgb, pn, counters);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// SDF::interpolate_distance (sdf.cpp:127-163) on the device layout.
// Returns false when no corner is valid (reference: is_interpolated = false, value NaN).
// `viol` is raised when a corner lies inside the grid but outside this rank's stored layers.

struct Vol {
    const float2* dw;
    int m, xs, xe;
};

__device__ __forceinline__ bool interp(const Vol& V, double vx, double vy, double vz, float& out, unsigned& viol) {
    const float fi = (float)vx, fj = (float)vy, fk = (float)vz;          // f64 -> f32, sdf.cpp:130-132
    const int bi = trunc_x86(fi), bj = trunc_x86(fj), bk = trunc_x86(fk);
    // issue all 8 corner loads first (independent), then run the reference's accumulation order
    float2 c[8];
    bool in[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ci = bi + (q >> 2), cj = bj + ((q >> 1) & 1), ck = bk + (q & 1);
        bool ok = (ci >= 0) & (cj >= 0) & (ck >= 0) & (ci < V.m) & (cj < V.m) & (ck < V.m);   // sdf.h:113-119
        if (ok && (ci < V.xs || ci >= V.xe)) { viol = 1u; ok = false; }
        in[q] = ok;
        const long long idx = ok ? ((long long)(ci - V.xs) * V.m + cj) * V.m + ck : 0ll;
        c[q] = ok ? V.dw[idx] : make_float2(0.0f, 0.0f);
    }
    float w_sum = 0.0f, sum_d = 0.0f;
    bool any = false;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ci = bi + (q >> 2), cj = bj + ((q >> 1) & 1), ck = bk + (q & 1);
        const float volume = (fabsf((float)ci - fi) + fabsf((float)cj - fj)) + fabsf((float)ck - fk);
        if (in[q] && c[q].y > 0.0f) {
            any = true;
            if ((double)volume < 0.00001) { out = c[q].x; return true; }   // exact hit, sdf.cpp:151-153
            const float w = 1.0f / volume;
            w_sum += w;
            sum_d += w * c[q].x;
        }
    }
    out = sum_d / w_sum;
    return any;
}

__global__ __launch_bounds__(256) void sample_kernel(Grid g, const float2* __restrict__ dw,
                                                      const double* __restrict__ vox, int n,
                                                      float* __restrict__ val, int* __restrict__ okv) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    Vol V{dw, g.m, g.xs, g.xe};
    float out = 0.0f;
    unsigned viol = 0;
    const bool ok = interp(V, vox[3 * t + 0], vox[3 * t + 1], vox[3 * t + 2], out, viol);
    val[t] = out;
    okv[t] = viol ? -1 : (ok ? 1 : 0);
}

hipError_t launch_sample(hipStream_t s, const Grid& g, const float2* dw, const double* vox, int32_t n,
                         float* val, int32_t* ok) {
    if (n <= 0) return hipSuccess;
    sample_kernel<<<dim3((n + 255) / 256), dim3(256), 0, s>>>(g, dw, vox, n, val, ok);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Tracker: one Gauss-Newton accumulation pass.
//
// One thread per sampled pixel, in the reference's visiting order (so a wavefront's ballot is a
// 64-sample window of that order).  Classification (NaN / out-of-grid / in-grid) needs geometry
// only; the reference's stale carry-over (an out-of-grid pixel re-adds the previous successful
// pixel's terms, camera_tracking.cpp:156-159,176-182,261-268) becomes a multiplicity
//     1 + #{out-of-grid samples between this sample and the next in-grid one, NaN samples skipped}
// computed from 64-bit ballots held in LDS, with a cooperative look-ahead past the workgroup's end.
// 13 look-ups x 8 corners of 8-byte {D,W} gathers per owned in-grid sample follow; the 27 unique
// terms of J J^T / r J (+ counters) are reduced wave-shuffle -> LDS across the 4 waves -> one row of
// `partials` per workgroup; track_final_kernel adds the rows in a fixed order (bitwise reproducible).

enum { kClsSkip = 0, kClsOog = 1, kClsIn = 2 };

struct SampleGeom {
    double px, py, pz;   // camera-frame point
    double vx, vy, vz;   // continuous voxel coordinates of its world position
};

__device__ __forceinline__ int classify(const TrackParams& p, const float4* __restrict__ samples, int n,
                                        SampleGeom& sg) {
    if (n >= p.n_samples) return kClsSkip;
    const float4 s = samples[n];
    if (is_nan(s.x) || is_nan(s.y) || is_nan(s.z)) return kClsSkip;          // camera_tracking.cpp:168
    sg.px = (double)s.x; sg.py = (double)s.y; sg.pz = (double)s.z;
    // project_camera_to_world (:55-58) + get_voxel_coordinates (sdf.h:143-147)
    const double wx = row3(&p.rot[0], sg.px, sg.py, sg.pz) + p.trans[0];
    const double wy = row3(&p.rot[3], sg.px, sg.py, sg.pz) + p.trans[1];
    const double wz = row3(&p.rot[6], sg.px, sg.py, sg.pz) + p.trans[2];
    sg.vx = (wx - p.g.origin[0]) * (double)p.g.m_div_w - 0.5;
    sg.vy = (wy - p.g.origin[1]) * (double)p.g.m_div_h - 0.5;
    sg.vz = (wz - p.g.origin[2]) * (double)p.g.m_div_d - 0.5;
    const double dm = (double)p.g.m;
    if (sg.vx < 0 || sg.vy < 0 || sg.vz < 0) return kClsOog;                 // :261-264
    if (sg.vx >= dm || sg.vy >= dm || sg.vz >= dm) return kClsOog;           // :265-268
    return kClsIn;
}

__device__ __forceinline__ void voxel_of(const TrackParams& p, const double* R, const SampleGeom& sg,
                                         double& vx, double& vy, double& vz) {
    const double wx = row3(&R[0], sg.px, sg.py, sg.pz) + p.trans[0];
    const double wy = row3(&R[3], sg.px, sg.py, sg.pz) + p.trans[1];
    const double wz = row3(&R[6], sg.px, sg.py, sg.pz) + p.trans[2];
    vx = (wx - p.g.origin[0]) * (double)p.g.m_div_w - 0.5;
    vy = (wy - p.g.origin[1]) * (double)p.g.m_div_h - 0.5;
    vz = (wz - p.g.origin[2]) * (double)p.g.m_div_d - 0.5;
}

__global__ __launch_bounds__(kTrackBlock) void track_kernel(TrackParams p, const float2* __restrict__ dw,
                                                             const float4* __restrict__ samples,
                                                             double* __restrict__ partials) {
    constexpr int NW = kTrackBlock / 64;
    __shared__ unsigned long long s_in[NW], s_oog[NW];     // this workgroup's windows
    __shared__ unsigned long long s_in2[NW], s_oog2[NW];   // look-ahead windows
    __shared__ double s_red[NW][kRedWidth];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = blockIdx.x * kTrackBlock + tid;

    SampleGeom sg;
    const int cls = classify(p, samples, n, sg);
    const unsigned long long b_in = __ballot(cls == kClsIn);
    const unsigned long long b_oog = __ballot(cls == kClsOog);
    if (lane == 0) { s_in[wv] = b_in; s_oog[wv] = b_oog; }
    __syncthreads();

    // ---- stale-carry multiplicity
    unsigned mult = 1;
    if (p.stale_carry) {
        // out-of-grid samples that follow this workgroup before the next in-grid sample
        unsigned tail = 0;
        bool any_in = false;
        for (int q = 0; q < NW; ++q) any_in |= (s_in[q] != 0ull);
        if (any_in) {
            bool found = false;
            for (int pos = (blockIdx.x + 1) * kTrackBlock; !found && pos < p.n_samples; pos += kTrackBlock) {
                SampleGeom tmp;
                const int c2 = classify(p, samples, pos + tid, tmp);
                const unsigned long long i2 = __ballot(c2 == kClsIn);
                const unsigned long long o2 = __ballot(c2 == kClsOog);
                __syncthreads();                        // previous round's readers are done
                if (lane == 0) { s_in2[wv] = i2; s_oog2[wv] = o2; }
                __syncthreads();
                for (int q = 0; q < NW && !found; ++q) {
                    const unsigned long long mi = s_in2[q], mo = s_oog2[q];
                    if (mi) {
                        const int nxt = __ffsll((long long)mi) - 1;
                        tail += __popcll(mo & ((1ull << nxt) - 1ull));
                        found = true;
                    } else {
                        tail += __popcll(mo);
                    }
                }
            }
        }
        if (cls == kClsIn) {
            const unsigned long long above = (lane == 63) ? 0ull : (~0ull << (lane + 1));
            unsigned cnt = 0;
            bool found = false;
            unsigned long long mi = s_in[wv] & above;
            if (mi) {
                const int nxt = __ffsll((long long)mi) - 1;
                cnt = __popcll(s_oog[wv] & above & ((1ull << nxt) - 1ull));
                found = true;
            } else {
                cnt = __popcll(s_oog[wv] & above);
                for (int q = wv + 1; q < NW && !found; ++q) {
                    mi = s_in[q];
                    if (mi) {
                        const int nxt = __ffsll((long long)mi) - 1;
                        cnt += __popcll(s_oog[q] & ((1ull << nxt) - 1ull));
                        found = true;
                    } else {
                        cnt += __popcll(s_oog[q]);
                    }
                }
                if (!found) cnt += tail;
            }
            mult = 1u + cnt;
        }
    }

    // ---- data association + numeric Jacobian for owned in-grid samples (camera_tracking.cpp:246-363)
    double acc[kRedWidth];
#pragma unroll
    for (int e = 0; e < kRedWidth; ++e) acc[e] = 0.0;
    acc[33] = (n < p.n_samples) ? 1.0 : 0.0;
    acc[32] = (n < p.n_samples && cls == kClsSkip) ? 1.0 : 0.0;
    acc[31] = (cls == kClsOog) ? 1.0 : 0.0;

    const bool owned = (cls == kClsIn) && (sg.vx >= (double)p.g.own_x0) && (sg.vx < (double)p.g.own_x1);
    if (owned) {
        acc[30] = 1.0;
        Vol V{dw, p.g.m, p.g.xs, p.g.xe};
        unsigned viol = 0;
        double J[6];
        float r0 = 0.0f, fp = 0.0f, fm = 0.0f;
        bool ok = interp(V, sg.vx, sg.vy, sg.vz, r0, viol);                     // :269
        // translation columns: +-v_h along each voxel axis                      :273-316
        if (ok) {
            const double vh = (double)p.v_h;
            ok = interp(V, sg.vx + vh, sg.vy, sg.vz, fp, viol) && interp(V, sg.vx - vh, sg.vy, sg.vz, fm, viol);
            J[0] = (double)((fp - fm) / p.vh2[0]);
            if (ok) {
                ok = interp(V, sg.vx, sg.vy + vh, sg.vz, fp, viol) && interp(V, sg.vx, sg.vy - vh, sg.vz, fm, viol);
                J[1] = (double)((fp - fm) / p.vh2[1]);
            }
            if (ok) {
                ok = interp(V, sg.vx, sg.vy, sg.vz + vh, fp, viol) && interp(V, sg.vx, sg.vy, sg.vz - vh, fm, viol);
                J[2] = (double)((fp - fm) / p.vh2[2]);
            }
        }
        // rotation columns: (I +- w_h [e_k]x) rot applied to the camera point   :318-361
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            if (ok) {
                double ax, ay, az, bx, by, bz;
                voxel_of(p, &p.rpm[9 * (2 * a + 0)], sg, ax, ay, az);
                voxel_of(p, &p.rpm[9 * (2 * a + 1)], sg, bx, by, bz);
                ok = interp(V, ax, ay, az, fp, viol) && interp(V, bx, by, bz, fm, viol);
                J[3 + a] = (double)((fp - fm) / p.wh2);
            }
        }
        acc[28] = viol ? 1.0 : 0.0;
        if (ok && !viol) {
            const double mu = (double)mult;
            const double r = (double)r0;
            int e = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a)
#pragma unroll
                for (int b = a; b < 6; ++b) acc[e++] = mu * (J[a] * J[b]);       // :181
#pragma unroll
            for (int a = 0; a < 6; ++a) acc[21 + a] = mu * (r * J[a]);           // :182
            acc[27] = mu;
            acc[29] = 1.0;
        }
    }

    // ---- reduction: wave butterfly, then LDS across the 4 waves, one row per workgroup
#pragma unroll
    for (int e = 0; e < kRedWidth; ++e) {
        double v = acc[e];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        acc[e] = v;
    }
    if (lane == 0) {
#pragma unroll
        for (int e = 0; e < kRedWidth; ++e) s_red[wv][e] = acc[e];
    }
    __syncthreads();
    if (tid < kRedWidth) {
        double v = s_red[0][tid];
        for (int q = 1; q < NW; ++q) v += s_red[q][tid];
        partials[(long long)blockIdx.x * kRedWidth + tid] = v;
    }
}

// Fixed-order final sum (kernel boundary = visibility; no atomics, no spin).
__global__ __launch_bounds__(256) void track_final_kernel(const double* __restrict__ partials, int nblocks,
                                                           double* __restrict__ red_dev,
                                                           double* __restrict__ red_host) {
    constexpr int ROWS = 256 / kRedWidth;            // 7 row groups of 34 columns
    __shared__ double s[ROWS][kRedWidth];
    const int tid = threadIdx.x;
    const int col = tid % kRedWidth, rg = tid / kRedWidth;
    if (rg < ROWS) {
        double v = 0.0;
        for (int b = rg; b < nblocks; b += ROWS) v += partials[(long long)b * kRedWidth + col];
        s[rg][col] = v;
    }
    __syncthreads();
    if (tid < kRedWidth) {
        double v = s[0][tid];
        for (int q = 1; q < ROWS; ++q) v += s[q][tid];
        red_dev[tid] = v;
        if (red_host) red_host[tid] = v;
    }
}

int track_num_blocks(int32_t n_samples) { return (n_samples + kTrackBlock - 1) / kTrackBlock; }

hipError_t launch_track(hipStream_t s, const TrackParams& p, const float2* dw, const float4* samples,
                        double* partials, double* red_dev, double* red_host) {
    const int nb = track_num_blocks(p.n_samples);
    if (nb <= 0) return hipErrorInvalidValue;
    track_kernel<<<dim3(nb), dim3(kTrackBlock), 0, s>>>(p, dw, samples, partials);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    track_final_kernel<<<dim3(1), dim3(256), 0, s>>>(partials, nb, red_dev, red_host);
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
